"""Fuzz the PRL front-end (pine_amd/host/prl.cpp parses untrusted text): mutate the example scripts and run each
mutant through pine_prl_interpret in dry-run mode (no GPU).  Every mutant must either run or fail with a PrlError;
under the sanitizer build (tools/sanitize/run.sh) any memory error or undefined behaviour aborts the process.

usage: python tools/fuzz_prl.py [mutants per script = 300] [seed = 1]"""
import glob
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pine_amd import prl  # noqa: E402

TOKENS = ["(", ")", "[", "]", "{", "}", ",", ";", ":=", "=", "+", "-", "*", "/", "^", "%", "..", "~", ".", "\"", "#", "\n",
          "for", "in", "while", "if", "else", "fn", "return", "true", "false", "1e39", "-0", "2147483648", "0x", "1.", ".5",
          "scene", "Rect", "Box", "Mesh", "Diffuse", "Emissive", "render", "save", "PathIntegrator", "BlueSampler", "Film"]


def mutate(src, r):
    s = src
    for _ in range(r.randint(1, 4)):
        k = r.randrange(8)
        i = r.randrange(len(s) + 1)
        j = min(len(s), i + r.randint(1, 12))
        if k == 0: s = s[:i] + s[j:]                                   # delete a span
        elif k == 1: s = s[:i] + s[i:j] + s[i:]                        # duplicate a span
        elif k == 2: s = s[:i] + r.choice(TOKENS) + s[i:]              # insert a token
        elif k == 3: s = s[:i] + r.choice(TOKENS) + s[j:]              # replace a span by a token
        elif k == 4: s = s[:i]                                         # truncate
        elif k == 5: s = s[:i] + "".join(chr(r.randrange(1, 256)) for _ in range(r.randint(1, 6))) + s[i:]  # raw bytes
        elif k == 6: s = s[:i] + str(r.choice([0, -1, 1 << 31, 10 ** 12, 1e308, -1e-308])) + s[j:]          # extreme numbers
        else:
            a, b = sorted((r.randrange(len(s) + 1), r.randrange(len(s) + 1)))
            s = s[:a] + s[b:] + s[a:b]                                 # move a block to the end
    return s.replace("\x00", " ")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    scripts = sorted(glob.glob(os.path.join(ROOT, "examples", "*.pine")))
    ran = failed = 0
    for path in scripts:
        src = open(path, encoding="utf-8").read()
        prl.interpret(src, dry_run=True)  # the unmutated script must run
        for _ in range(n):
            m = mutate(src, r)
            try:
                prl.interpret(m.encode("utf-8", "replace").decode("utf-8", "replace"), dry_run=True)
                ran += 1
            except prl.PrlError:
                failed += 1
    print(f"fuzz_prl: {len(scripts)} scripts x {n} mutants: {ran} ran, {failed} rejected with an error, 0 crashes")


if __name__ == "__main__":
    main()
