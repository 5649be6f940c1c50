"""Hash of the library's sources (kernels + C ABI + mirror headers): printed as 12 hex digits.  bench.py compares it with the hash
recorded next to a counter summary in profiles/ and drops `roofline.traffic` when the summary was taken on other code (there is
no .git on a GPU box to ask for the commit)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for d in ("fast_lio_amd/csrc", "include", "include/fastlio_amd"):
        p = os.path.join(root, d)
        for f in sorted(os.listdir(p)):
            fp = os.path.join(p, f)
            if os.path.isfile(fp) and f.rsplit(".", 1)[-1] in ("hip", "inc", "hpp", "cpp", "h"):
                h.update(f.encode())
                h.update(open(fp, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(src_hash())
