"""Writes the BMP fixtures (own data): one small file per case Bitmap::LoadBMP (Core/Utils/BitmapBMP.cpp:30-140) distinguishes --
24-bit, 8-bit without a colour table (R8), 8-bit with a colour table of 2 / 16 / 256 entries (B8G8R8A8_UNorm_Palette), widths that
need row padding, and files it refuses (4-bit, RLE-compressed, truncated).  Regenerate: python tests/golden/bmp/make_bmp_fixture.py"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def bmp(width, height, bits, palette_entries=0, compression=0, planes=1, truncate=0, seed=1):
    stride = (width * bits // 8 + 3) & ~3
    palette = bytes(((seed * 37 + 11 * i + 3 * c) & 255) for i in range(palette_entries) for c in range(4))
    data = bytes(((seed * 17 + 7 * i) % (palette_entries if palette_entries else 256)) for i in range(stride * height))
    off = 14 + 40 + len(palette)
    info = struct.pack("<IiiHHIIiiII", 40, width, height, planes, bits, compression, len(data), 2835, 2835, palette_entries, 0)
    head = struct.pack("<2sIHHI", b"BM", off + len(data), 0, 0, off)
    blob = head + info + palette + data
    return blob[:len(blob) - truncate] if truncate else blob


CASES = [
    ("00_rgb24_8x8", dict(width=8, height=8, bits=24)),
    ("01_rgb24_5x3_padded", dict(width=5, height=3, bits=24, seed=2)),
    ("02_gray8_8x8", dict(width=8, height=8, bits=8, seed=3)),
    ("03_gray8_7x5_padded", dict(width=7, height=5, bits=8, seed=4)),
    ("04_pal8_16_colours", dict(width=8, height=8, bits=8, palette_entries=16, seed=5)),
    ("05_pal8_256_colours", dict(width=16, height=4, bits=8, palette_entries=256, seed=6)),
    ("06_pal8_2_colours_padded", dict(width=6, height=6, bits=8, palette_entries=2, seed=7)),
    ("07_pal4_refused", dict(width=8, height=8, bits=4, palette_entries=16, seed=8)),
    ("08_rle8_refused", dict(width=8, height=8, bits=8, palette_entries=16, compression=1, seed=9)),
    ("09_rgb24_truncated", dict(width=8, height=8, bits=24, truncate=40, seed=10)),
    ("10_planes2_refused", dict(width=8, height=8, bits=24, planes=2, seed=11)),
]

if __name__ == "__main__":
    for name, kw in CASES:
        with open(os.path.join(HERE, name + ".bmp"), "wb") as f:
            f.write(bmp(**kw))
    print("wrote %d files" % len(CASES))
