#!/usr/bin/env python3
"""Pack the BlueSobol sampler's numeric tables into one u8 blob.

The tables are published data (Heitz et al. 2019, "A Low-Discrepancy Sampler that Distributes
Monte Carlo Errors as a Blue Noise in Screen Space"); the reference carries them as C arrays in
/root/reference/src/contrib/bluesobol/bluenoise_{1..256}spp.cpp:2,7,12.  The sampler is a table
lookup, so bit-exact parity needs the same numbers.  This script parses the integer
initialisers (data only -- no code is copied) and writes

    pine_amd/data/bluesobol_u8.bin =
        sobol_256spp_256d  u8[256*256]                       (identical in all nine files; checked)
        for spp in 1,2,4,...,256:  scramblingTile u8[128*128*8], rankingTile u8[128*128*8]

All values are < 256 (checked), so u8 storage is lossless.  Run in the build container only.
"""
import re, sys, pathlib
import numpy as np

REF = pathlib.Path("/root/reference/src/contrib/bluesobol")
OUT = pathlib.Path(__file__).resolve().parent.parent / "pine_amd" / "data" / "bluesobol_u8.bin"

def arrays(path):
    text = path.read_text()
    out = {}
    for m in re.finditer(r"(?:static|extern) const int (\w+)\[[^\]]*\]\s*=\s*\{([^}]*)\}", text):
        vals = np.array(re.findall(r"-?\d+", m.group(2)), dtype=np.int64)
        out[m.group(1)] = vals
    return out

def main():
    blob = []
    sobol = None
    for k in range(9):
        spp = 1 << k
        a = arrays(REF / f"bluenoise_{spp}spp.cpp")
        s, sc, rk = a["sobol_256spp_256d"], a["scramblingTile"], a["rankingTile"]
        assert s.size == 65536 and sc.size == 131072 and rk.size == 131072, (spp, s.size, sc.size, rk.size)
        for t in (s, sc, rk):
            assert t.min() >= 0 and t.max() < 256
        if sobol is None:
            sobol = s
            blob.append(s.astype(np.uint8))
        else:
            assert np.array_equal(sobol, s), f"sobol table differs in {spp}spp"
        blob.append(sc.astype(np.uint8))
        blob.append(rk.astype(np.uint8))
    data = np.concatenate(blob)
    assert data.size == 65536 + 9 * 262144
    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_bytes(data.tobytes())
    print("wrote", OUT, data.size, "bytes")

if __name__ == "__main__":
    main()
