"""Host-only checks of zig/hip_backend.zig (no Zig toolchain in this image, so the file cannot be compiled here).

1. Every name it takes from the reference -- `rt.X` (raytracer.zig) and `img_io.X` (image.zig) -- must be declared
   `pub` there: Zig rejects a private declaration across files, and the shim is meant to drop into the reference's
   src/ with the reference's files unmodified (boundary defect: rt.Sphere and rt.shape_s are private).
   The pub list is tests/golden/reference_decls.json, transcribed from the reference by make_reference_decls.py.
2. Every `extern fn` it declares must exist in include/rt_hip.h with the same number of parameters, and the extern
   structs must list the header's fields in the header's order.
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZIG = open(os.path.join(ROOT, "zig", "hip_backend.zig")).read()
HDR = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
DECLS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_decls.json")))["files"]


def _code(src):
    """the file without comments and string literals"""
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r'"(\\.|[^"\\])*"', '""', src)


def test_reference_names_used_by_the_shim_are_pub():
    code = _code(ZIG)
    for prefix, ref_file in (("rt", "raytracer.zig"), ("img_io", "image.zig")):
        table = {d["name"]: d for d in DECLS[ref_file]}
        used = sorted(set(re.findall(r"\b%s\.(\w+)" % prefix, code)))
        assert used, prefix
        for name in used:
            assert name in table, "%s.%s is not a top-level declaration of %s" % (prefix, name, ref_file)
            assert table[name]["pub"], "%s.%s is private (%s:%d): the shim would not compile against the unmodified reference" \
                % (prefix, name, ref_file, table[name]["line"])
    # the two private types the previous round's shim named
    t = {d["name"]: d for d in DECLS["raytracer.zig"]}
    assert not t["Sphere"]["pub"] and not t["shape_s"]["pub"]
    assert "rt.Sphere" not in code and "rt.shape_s" not in code


def _c_params(sig):
    sig = sig.strip()
    if sig in ("", "void"):
        return 0
    return len(sig.split(","))


def test_extern_fns_match_the_header():
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    c_fns = {m.group(1): _c_params(m.group(2)) for m in re.finditer(r"\b(rt_\w+)\s*\(([^;{}]*?)\)\s*;", hdr)}
    zig_fns = re.findall(r"extern fn (rt_\w+)\((.*?)\) [\w\[\]:*? ]+;", _code(ZIG))
    assert len(zig_fns) >= 10
    for name, params in zig_fns:
        assert name in c_fns, "%s is not declared in include/rt_hip.h" % name
        n = 0 if not params.strip() else len(params.split(","))
        assert n == c_fns[name], (name, n, c_fns[name])


def _c_struct_fields(name):
    m = re.search(r"typedef struct \{([^{}]*)\} %s;" % name, re.sub(r"/\*.*?\*/", "", HDR, flags=re.S), flags=re.S)
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.sub(r"\[.*?\]", "", part).replace("*", " ").split()[-1])
    return fields


def _zig_struct_fields(name):
    m = re.search(r"pub const %s = extern struct \{([^{}]*)\};" % name, _code(ZIG), flags=re.S)
    return re.findall(r"(\w+)\s*:", m.group(1))


def test_extern_structs_list_the_headers_fields_in_order():
    for name in ("rt_camera", "rt_spheres", "rt_triangles", "rt_config", "rt_stats"):
        assert _zig_struct_fields(name) == _c_struct_fields(name), name
