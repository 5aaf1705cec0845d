"""Transcribes which top-level declarations of the reference's raytracer.zig / image.zig are `pub` (name, line,
kind -- no source text) into reference_decls.json.  Run in the build container (reads /root/reference)."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"

out = {}
for f in ("raytracer.zig", "image.zig"):
    decls = []
    with open(os.path.join(REF, f)) as fh:
        for i, line in enumerate(fh, 1):
            m = re.match(r"^(pub )?(const|var|fn) (\w+)", line)
            if m:
                decls.append({"name": m.group(3), "line": i, "pub": bool(m.group(1)), "kind": m.group(2)})
    out[f] = decls
note = ("Top-level declarations of the reference's src/raytracer.zig and src/image.zig: name, line, kind and whether "
        "declared `pub` (transcribed by tests/golden/make_reference_decls.py; names and line numbers only, no source "
        "text). Used by tests/test_zig_shim.py.")
with open(os.path.join(HERE, "reference_decls.json"), "w") as fh:
    json.dump({"note": note, "files": out}, fh, indent=1)
