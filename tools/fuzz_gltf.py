"""Fuzz the C++ glTF importer (pine_amd/host/gltf_import.hpp parses untrusted binary files): mutate tests/golden/import_test.glb
-- bytes of the JSON chunk (structure, indices, counts, offsets, strides), of the binary chunk, of the GLB header -- and run every
mutant through `load(scene, file)` of the PRL front-end in dry-run mode (no GPU).  A mutant must import or fail with an error;
under the sanitizer build (tools/sanitize/run.sh) any out-of-bounds read or undefined behaviour aborts the process.

usage: python tools/fuzz_gltf.py [mutants = 400] [seed = 1]"""
import os
import random
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pine_amd import prl  # noqa: E402

GLB = os.path.join(ROOT, "tests", "golden", "import_test.glb")
NUMBERS = [b"0", b"-1", b"1", b"65535", b"65536", b"2147483647", b"4294967295", b"99999999999", b"1e30", b"-0.5", b"3", b"12", b"36"]


def chunks(data):
    pos, out = 12, []
    while pos + 8 <= len(data):
        n, kind = struct.unpack("<II", data[pos:pos + 8])
        out.append((kind, pos + 8, n))
        pos += 8 + n
    return out


def mutate(data, r):
    b = bytearray(data)
    (_, jpos, jn), (_, bpos, bn) = chunks(data)[:2]
    for _ in range(r.randint(1, 3)):
        k = r.randrange(7)
        if k == 0:    # replace a number inside the JSON chunk, keeping the chunk length (pad / cut with spaces)
            import re
            js = bytes(b[jpos:jpos + jn])
            ms = list(re.finditer(rb"-?\d+(\.\d+)?", js))
            if ms:
                m = r.choice(ms)
                rep = r.choice(NUMBERS)
                new = js[:m.start()] + rep + js[m.end():]
                new = (new + b" " * jn)[:jn] if len(new) <= jn else new[:jn]
                b[jpos:jpos + jn] = new
        elif k == 1:  # flip bytes in the JSON chunk
            for _ in range(r.randint(1, 4)):
                b[jpos + r.randrange(jn)] = r.randrange(32, 127)
        elif k == 2:  # flip bytes in the binary chunk
            for _ in range(r.randint(1, 8)):
                b[bpos + r.randrange(bn)] = r.randrange(256)
        elif k == 3:  # corrupt a chunk header / the file header
            off = r.choice([8, 12, 16, bpos - 8, bpos - 4])
            b[off:off + 4] = struct.pack("<I", r.choice([0, 1, 7, len(b), len(b) + 1, 0xffffffff, r.randrange(1 << 32)]))
        elif k == 4:  # truncate
            del b[r.randrange(12, len(b)):]
            (_, jpos, jn), (_, bpos, bn) = (chunks(bytes(b)) + [(0, 12, 1), (0, 12, 1)])[:2]
            jn, bn = max(1, min(jn, len(b) - jpos)), max(1, min(bn, len(b) - bpos))
            if len(b) < 24:
                break
        elif k == 5:  # delete a span of the JSON (unbalanced brackets, cut strings)
            i = jpos + r.randrange(jn)
            j = min(jpos + jn, i + r.randint(1, 40))
            b[i:j] = b" " * (j - i)
        else:         # swap two spans of the JSON
            i, j = jpos + r.randrange(jn), jpos + r.randrange(jn)
            n = r.randint(1, 16)
            if i + n <= jpos + jn and j + n <= jpos + jn:
                b[i:i + n], b[j:j + n] = b[j:j + n], b[i:i + n]
    return bytes(b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    data = open(GLB, "rb").read()
    ok = bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "m.glb")
        for _ in range(n):
            open(path, "wb").write(mutate(data, r))
            try:
                prl.interpret(f'scene := Scene(); load(scene, "{path}");', dry_run=True)
                ok += 1
            except prl.PrlError:
                bad += 1
    print(f"glTF fuzz: {n} mutants, {ok} imported, {bad} refused, no crash")


if __name__ == "__main__":
    main()
