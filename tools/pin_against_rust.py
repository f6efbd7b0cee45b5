#!/usr/bin/env python3
"""PIN KIT: one command for anyone who has `cargo` and a checkout of bitshifter/pathtrace-rs.

    python3 tools/pin_against_rust.py /path/to/pathtrace-rs [--only small] [--no-build]

For every case of tests/golden/rust_expectations.json it runs `cargo run --release -- -O -P <preset> -W .. -H .. -S .. [-B]` in the
Rust checkout and compares
  (a) the "{}rays" count the binary prints (offline.rs:36-41) -- equality of, e.g., 162 554 454 rays on random_spheres 1200x800x64 pins
      every control-affecting assumption of the oracle at once (RNG stream and draw order, glam's normalize, sphere / scatter
      arithmetic, the BVH's accept rule),
  (b) the SHA-256 of the decoded RGB8 pixels of output.png (offline.rs:43-59),
  (c) when the checkout carries tools/pin/offline_dump_f32.patch (git apply it first): the SHA-256 of the raw f32 frame buffer.
Needs nothing but Python 3 (the PNG is decoded here, with zlib). Until this has been run once, parity of this repository is against
the oracle's RESTATEMENT of the reference, not against the reference binary."""
import argparse
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def png_rgb8(path):
    """Decoded pixels of an 8-bit RGB (or RGBA / grey) PNG as bytes, rows top-down, RGB only."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0 and ctype in (2, 6), "expected non-interlaced 8-bit RGB(A)"
            bpp = 3 if ctype == 2 else 4
        elif kind == b"IDAT":
            idat += body
    raw = zlib.decompress(idat)
    stride = w * bpp
    out, prev = bytearray(), bytearray(stride)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = bytearray(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)])
        for i in range(stride):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 1: line[i] = (line[i] + a) & 255
            elif f == 2: line[i] = (line[i] + b) & 255
            elif f == 3: line[i] = (line[i] + ((a + b) >> 1)) & 255
            elif f == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                line[i] = (line[i] + (a if pa <= pb and pa <= pc else (b if pb <= pc else c))) & 255
        prev = line
        out += line if bpp == 3 else bytes(v for i, v in enumerate(line) if i % 4 != 3)
    return bytes(out)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkout", help="path of a bitshifter/pathtrace-rs checkout")
    ap.add_argument("--only", default=None, help="run the cases of one preset")
    ap.add_argument("--no-build", action="store_true", help="use target/release/pathtrace-rs as it is")
    a = ap.parse_args()
    exp = json.load(open(os.path.join(os.path.dirname(HERE), "tests", "golden", "rust_expectations.json")))
    if not a.no_build:
        subprocess.check_call(["cargo", "build", "--release"], cwd=a.checkout)
    failures = 0
    for c in exp["cases"]:
        if a.only and a.only not in c["args"]:
            continue
        env = dict(os.environ, PT_DUMP_F32=os.path.join(a.checkout, "output.f32"))
        for f in ("output.png", "output.f32"):
            if os.path.exists(os.path.join(a.checkout, f)):
                os.remove(os.path.join(a.checkout, f))
        out = subprocess.run(["cargo", "run", "--release", "--"] + c["args"], cwd=a.checkout, env=env, capture_output=True, text=True)
        m = re.search(r"(\d+)rays", out.stdout)
        rays = int(m.group(1)) if m else None
        png = hashlib.sha256(png_rgb8(os.path.join(a.checkout, "output.png"))).hexdigest() if os.path.exists(os.path.join(a.checkout, "output.png")) else None
        f32p = os.path.join(a.checkout, "output.f32")
        f32 = hashlib.sha256(open(f32p, "rb").read()).hexdigest() if os.path.exists(f32p) else None
        ok = rays == c["rays"] and png == c["rgb8_sha256"] and (f32 is None or f32 == c["f32_sha256"])
        failures += 0 if ok else 1
        print("%-4s %-58s rays %s%s  png %s  f32 %s%s" % (
            "ok" if ok else "DIFF", " ".join(c["args"]), rays, "" if rays == c["rays"] else " (expected %d)" % c["rays"],
            "=" if png == c["rgb8_sha256"] else "DIFFERS", "=" if f32 == c["f32_sha256"] else ("not dumped (apply tools/pin/offline_dump_f32.patch)" if f32 is None else "DIFFERS"),
            "   [colour passes through libm here: a different libm may move a last bit]" if (not ok and c.get("libm_sensitive")) else ""))
    print("%d case(s) differ" % failures if failures else "all cases equal: the oracle is pinned to this Rust binary")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
