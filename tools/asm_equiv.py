#!/usr/bin/env python
"""Is the DEVICE code of the product library the same, kernel for kernel, as at another commit?  (No GPU needed.)

    python tools/asm_equiv.py <commit> [--bless]  # e.g. the commit the profiles under profiles/ were taken on

Checks out <commit>'s fast_lio_amd/csrc + include into a scratch directory, compiles every .hip source of both trees to gfx950
assembly with the product's flags (fast_lio_amd/_build.py) and compares the kernels' instruction streams (tools/asm_same.py's
comparison: comments, directives and block-label numbers ignored).  Used to show that experiment code added under
#ifdef FLH_EXP_* (developer builds, tools/variant.py) leaves the product's machine code untouched, so that the counter summaries
taken on <commit> still describe the library at HEAD.  Prints one line per source and exits 1 on any difference."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_lio_amd import _build  # noqa: E402


def bodies(path, prefix="_ZN3flh"):
    lines = open(path).read().split("\n")
    res = {}
    for i, l in enumerate(lines):
        if l.startswith(prefix) and "@_ZN3flh" in l:
            name = l.split(":")[0]
            j = i + 1
            out = []
            while not lines[j].startswith(".Lfunc_end"):
                s = lines[j].strip()
                j += 1
                if not s or s.startswith((";", ".")):
                    continue
                out.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", s).strip()))
            res[name] = out
    return res


def asm(tree, src, out):
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([_build.hipcc()] + flags + ["-I", os.path.join(tree, "include"), "--cuda-device-only", "-S", "-x", "hip",
                                                      os.path.join(tree, "fast_lio_amd", "csrc", src), "-o", out])


def main():
    refs = [a for a in sys.argv[1:] if not a.startswith("--")]
    ref = refs[0] if refs else "HEAD"
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        old = os.path.join(td, "old")
        os.makedirs(old)
        tar = subprocess.Popen(["git", "-C", ROOT, "archive", ref, "fast_lio_amd/csrc", "include"], stdout=subprocess.PIPE)
        subprocess.check_call(["tar", "-x", "-C", old], stdin=tar.stdout)
        tar.wait()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from src_hash import src_hash

        old_hash = src_hash(old)
        for src in _build.SOURCES:
            if not src.endswith(".hip"):
                continue
            a_s, b_s = os.path.join(td, "a.s"), os.path.join(td, "b.s")
            asm(old, src, a_s)
            asm(ROOT, src, b_s)
            a, b = bodies(a_s), bodies(b_s)
            diff = [k for k in set(a) | set(b) if a.get(k) != b.get(k)]
            print(f"{src:20s} kernels at {ref}: {len(a):3d}  now: {len(b):3d}  differing: {len(diff)}")
            for k in diff[:5]:
                print("    ", k[:110])
            bad += len(diff)
    print("device code", "IDENTICAL" if bad == 0 else "DIFFERENT", "to", ref, "(sources", old_hash + ")")
    if bad == 0 and "--bless" in sys.argv:
        # the counter summaries taken on <ref> describe today's kernels too: note today's source hash next to them (bench.py reads it)
        import glob
        import json

        here = src_hash()
        for mp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary_config*.meta.json"))):
            meta = json.load(open(mp))
            if meta.get("src_hash") != old_hash:  # only summaries taken on exactly the sources that were compared
                continue
            lst = meta.setdefault("device_code_identical_src_hashes", [])
            if here != meta.get("src_hash") and here not in lst:
                lst.append(here)
                json.dump(meta, open(mp, "w"))
                print("blessed", os.path.relpath(mp, ROOT), "for sources", here)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
