"""The exact update's MFMA tile by itself (`relp_debug_exact_tile_bench`): microseconds per tile per wave and word products per second of the
whole chip, by width of the entries and number of terms.  With RELP_AMD_LIB pointing at a build of exact.hip with -DRELP_TILE_VARIANT=n the same
with one resource taken out (1 no stores, 2 no LDS fragments, 3 no entries from memory, 4 no MFMAs, 5 no epilogue): what the tile waits for.

    python tools/tile_bench.py [limbs]          (default 128)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402


def main():
    limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    fn = relp_amd.lib().relp_debug_exact_tile_bench
    fn.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_double)]
    print("library %s, %d limbs, 512 workgroups x 4 waves" % (os.path.basename(relp_amd.api.LIB_PATH), limbs))
    for terms in (2, 1):
        for blocks in sorted({max(1, limbs // 32), max(1, limbs // 16), limbs // 8}):
            tiles = max(2, 4096 // (blocks * blocks * terms))
            seconds = C.c_double()
            status = fn(0, limbs, tiles, blocks, terms, 200, C.byref(seconds))
            assert status == 0, status
            per_tile = seconds.value / tiles
            products = 2048 * tiles * 16 * terms * (8 * blocks) * (8 * blocks + 1) / 2  # needed word products: 16 entries x terms x W (W + 1) / 2
            print("  %d term(s), %2d blocks of 64 bytes: %7.2f us per tile per wave, %6.2f T word products/s (peak 39.06)" % (
                terms, blocks, per_tile * 1e6, products / seconds.value / 1e12))


if __name__ == "__main__":
    main()
