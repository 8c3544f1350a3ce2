"""patch_expect.py — the build's two rewriting passes (irpatch.py on LLVM IR text, asmpatch.py on object code) are validated against
ONE toolchain (ROCm 7.2 / clang 22).  A compiler that shapes the IR or the function prologues differently could make a
regular expression match fewer sites than it should — silently: the library would still build, with some handlers spending
an extra v_readfirstlane pair per dispatch, or worse, waiting where they must not.  So every module's match counts are
recorded in csrc/patch_expect/<module>.json and every build compares: a difference FAILS the build.  After an intentional
change of the handler set, `DE_UPDATE_PATCH_EXPECT=1 bash build.sh` records the new counts (commit them)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def check(module, stats):
    path = os.path.join(HERE, "patch_expect", module + ".json")
    have = {}
    if os.path.exists(path):
        with open(path) as fh:
            have = json.load(fh)
    if os.environ.get("DE_PATCH_EXPECT") == "skip":  # a variant build for an A/B run (other -D flags: other counts)
        return
    if os.environ.get("DE_UPDATE_PATCH_EXPECT") == "1":
        have.update(stats)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(have, fh, indent=1, sort_keys=True)
        return
    bad = {k: (have.get(k), v) for k, v in stats.items() if have.get(k) != v}
    if bad:
        sys.exit(f"patch_expect: module {module}: match counts differ from csrc/patch_expect/{module}.json (expected, got): {bad}\n"
                 "  a different compiler or an unintended change of the handler set: the IR / object patches may have missed sites.\n"
                 "  If the change is intentional: DE_UPDATE_PATCH_EXPECT=1 bash build.sh, then commit csrc/patch_expect/.")
