import sys,re
for line in sys.stdin:
    if 'conv_i8' not in line: continue
    name=line[:46].strip(); d={k:float(v) for k,v in re.findall(r'(\w+)=([0-9.e+]+)\(', line)}
    n=7; waves=d['SQ_WAVES']/n
    wc=d['SQ_WAVE_CYCLES']
    cyc=d['SQ_BUSY_CYCLES']/n/32
    print(name, 'waves %.0f'%waves, 'kernel cycles %.0f'%cyc, 'mfma busy %.3f'%(d['SQ_VALU_MFMA_BUSY_CYCLES']/n/(1024*cyc)))
    print('   wave time: parked %.2f issue-stalled %.2f issuing %.2f (valu %.2f sca %.2f lds %.2f vmem %.2f)'%(d['SQ_WAIT_ANY']/wc,d['SQ_WAIT_INST_ANY']/wc,d['SQ_ACTIVE_INST_ANY']/wc,d['SQ_ACTIVE_INST_VALU']/wc,d['SQ_ACTIVE_INST_SCA']/wc,d['SQ_ACTIVE_INST_LDS']/wc,d['SQ_ACTIVE_INST_VMEM']/wc))
    print('   per wave: VALU %.0f SALU %.0f LDS %.0f MFMA %.0f VMEM %.0f | LDS active/cycle/CU %.2f conflicts/active %.2f'%(d['SQ_INSTS_VALU']/n/waves,d['SQ_INSTS_SALU']/n/waves,d['SQ_INSTS_LDS']/n/waves,d['SQ_INSTS_MFMA']/n/waves,d['SQ_INSTS_VMEM']/n/waves,d['SQ_LDS_IDX_ACTIVE']/n/256/cyc,d['SQ_LDS_BANK_CONFLICT']/d['SQ_LDS_IDX_ACTIVE']))
