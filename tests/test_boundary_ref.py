"""The drop-in boundary, pinned by the reference's OWN callers and headers (CPU, build container only: skipped where
/root/reference is absent, e.g. on the GPU box).

* The reference's programs that drive this path -- src/mars/mars_test.c, src/mars/mars_yolo_test.c, examples/test_init.c,
  examples/test_inference.c -- are compiled UNCHANGED, from where they lie, against this repo's include/ and linked with
  thingino-accel_amd/lib/libnna_mars.so (third-party stb headers come from the reference tree's vendored copy, after ours
  on the search path).  Nothing is copied; the binaries land in a temporary directory.
* A layout probe generated from THIS repo's headers (every struct's size, every field's offset and size, every
  enumerator's value) is compiled once per header set; the two outputs must be identical.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
REF = "/root/reference"
LIB = os.path.join(ROOT, "thingino-accel_amd", "lib")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "mars")), reason="reference tree absent")

CALLERS = ["src/mars/mars_test.c", "src/mars/mars_yolo_test.c", "examples/test_init.c", "examples/test_inference.c"]


@pytest.mark.parametrize("rel", CALLERS)
def test_reference_caller_builds_and_links_unchanged(rel, tmp_path):
    exe = tmp_path / os.path.basename(rel)[:-2]
    # our headers first; the reference's include/ only AFTER them (for the vendored third-party stb/ files)
    # -D_GNU_SOURCE: examples/test_inference.c uses Dl_info, which glibc (unlike the camera's libc) only declares under it
    cmd = ["gcc", "-O1", "-w", "-D_GNU_SOURCE", "-I", INC, "-idirafter", os.path.join(REF, "include"), os.path.join(REF, rel),
           "-L", LIB, "-lnna_mars", "-Wl,-rpath," + LIB, "-lm", "-ldl", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # every nna_* / mars_* symbol the program imports is one libnna_mars.so defines
    und = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    need = {l.split()[-1].split("@")[0] for l in und.splitlines() if re.search(r"\b(nna_|mars_|mxu_)", l)}
    have = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(LIB, "libnna_mars.so")], text=True)
    have = {l.split()[-1] for l in have.splitlines()}
    assert need and need <= have, sorted(need - have)
    # the dependency the preprocessor resolved for the API headers is ours, not the reference's
    deps = subprocess.check_output(["gcc", "-w", "-D_GNU_SOURCE", "-M", "-I", INC, "-idirafter", os.path.join(REF, "include"), os.path.join(REF, rel)], text=True)
    for h in ("nna.h", "mars_runtime.h", "mars.h", "nna_model.h"):
        for path in re.findall(r"(\S*/%s)\b" % re.escape(h), deps):
            assert path.startswith(INC), path


def _fields(body, prefix=""):
    """field paths of a struct / union body; anonymous-typed nested `union { ... } name;` members are walked into"""
    out, i, depth, cur = [], 0, 0, ""
    while i < len(body):
        ch = body[i]
        if ch == "{":
            j, d = i, 0
            while True:  # matching brace
                d += body[j] == "{"
                d -= body[j] == "}"
                if d == 0:
                    break
                j += 1
            k = body.index(";", j)
            name = body[j + 1:k].strip()
            out.append(prefix + name)
            out += _fields(body[i + 1:j], prefix + name + ".")
            i, cur = k + 1, ""
            continue
        if ch == ";":
            for piece in cur.split(","):
                fm = re.search(r"(\w+)\s*(?:\[[^\]]*\]\s*)*$", piece.strip())
                if fm:
                    out.append(prefix + fm.group(1))
            cur = ""
        else:
            cur += ch
        i += 1
    return out


def _probe_source():
    """offsetof / sizeof of every struct field and the value of every enumerator declared in include/{mars,mars_runtime,
    nna_types,nna_model}.h (the headers shared with the reference), as a C program"""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mars.h"', '#include "mars_runtime.h"', '#include "nna.h"',
             '#include "nna_types.h"', '#include "nna_memory.h"', '#include "nna_tensor.h"', '#include "nna_model.h"', '#include "mxu_ops.h"',
             '#define F(T, f) printf(#T "." #f " %zu %zu\\n", offsetof(T, f), sizeof(((T *)0)->f))', 'int main(void) {']
    n_fields = n_enums = 0
    for hdr in ("mars.h", "mars_runtime.h", "nna_types.h", "nna_model.h"):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, hdr)).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"typedef\s+struct\s*(?:__attribute__\s*\(\(packed\)\))?\s*\w*\s*\{", text):
            j, d = m.end() - 1, 0
            while True:
                d += text[j] == "{"
                d -= text[j] == "}"
                if d == 0:
                    break
                j += 1
            name = re.match(r"\s*(\w+)\s*;", text[j + 1:]).group(1)
            lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
            for f in _fields(text[m.end():j]):
                lines.append("F(%s, %s);" % (name, f))
                n_fields += 1
        for m in re.finditer(r"typedef\s+enum\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            for item in m.group(1).split(","):
                em = re.match(r"\s*(\w+)", item)
                if em:
                    lines.append('printf("%s %%d\\n", (int)%s);' % (em.group(1), em.group(1)))
                    n_enums += 1
    lines += ["return 0; }"]
    assert n_fields > 90 and n_enums > 50, (n_fields, n_enums)
    return "\n".join(lines)


def test_struct_layouts_and_enums_equal_the_reference_headers(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text(_probe_source())
    outs = []
    for inc in (INC, os.path.join(REF, "include")):
        exe = tmp_path / ("probe_" + ("ours" if inc == INC else "ref"))
        r = subprocess.run(["gcc", "-std=gnu11", "-w", "-I", inc, str(src), "-o", str(exe)], capture_output=True, text=True)
        assert r.returncode == 0, "%s: %s" % (inc, r.stderr[-3000:])
        outs.append(subprocess.check_output([str(exe)], text=True))
    assert outs[0] == outs[1]
    assert "mars_header_t 76\n" in outs[0] and "mars_tensor_t 124\n" in outs[0] and "mars_layer_t 112\n" in outs[0]


def test_function_prototypes_equal_the_reference_headers(tmp_path):
    """every function the reference's public headers for this path declare is declared here with the same type: a TU that
    includes OUR header and then re-declares each function with the REFERENCE's prototype text must compile (C rejects
    conflicting redeclarations)"""
    decls = []
    for hdr in ("nna.h", "nna_memory.h", "nna_tensor.h", "mars_runtime.h", "mxu_ops.h", "nna_model.h"):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REF, "include", hdr)).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"^[ \t]*((?:const\s+)?[\w]+[\w\s\*]*?\b(?:nna_|mars_|mxu_|conv2d_)\w+\s*\([^;{]*\))\s*;", text, flags=re.M):
            if "static" not in m.group(1) and "typedef" not in m.group(1):
                decls.append(m.group(1))
    assert len(decls) >= 45, len(decls)
    src = tmp_path / "protos.c"
    src.write_text("\n".join(['#include "nna.h"', '#include "nna_memory.h"', '#include "nna_tensor.h"', '#include "mars_runtime.h"',
                              '#include "mxu_ops.h"', '#include "nna_model.h"'] + [d + ";" for d in decls] + ["int main(void){return 0;}"]))
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", INC, "-c", str(src), "-o", str(tmp_path / "p.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
