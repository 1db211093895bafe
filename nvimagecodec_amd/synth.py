"""Deterministic, integer-only synthetic "photo-like" RGB images.

A coarse random colour grid is bilinearly upsampled in fixed point (low-frequency content) and a small
approximately-Gaussian noise term (sigma ~ 6) is added.  Everything is integer arithmetic on a counter-based
hash, so the same (width, height, seed) yields the same bytes on every machine -- the bench, the tests and the
CPU baseline all decode identical inputs.  At quality 90 / 4:2:0 a 1920x1080 image compresses to ~0.5 MB.
"""
import numpy as np


def _hash_u32(x):
    """splitmix-style avalanche on uint64 counters -> uint32"""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = x ^ (x >> np.uint64(31))
    return (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def synth_image(width, height, seed=0, noise=True):
    """-> uint8 array [height, width, 3] (RGB)"""
    gw, gh = 17, 10  # coarse grid nodes
    with np.errstate(over="ignore"):
        node_ctr = (np.arange(gw * gh * 3, dtype=np.uint64) + np.uint64(seed) * np.uint64(1000003))
        nodes = (_hash_u32(node_ctr) >> np.uint32(24)).astype(np.int64).reshape(gh, gw, 3)  # 0..255
        # fixed-point (16.16) sample positions
        fx = (np.arange(width, dtype=np.int64) * ((gw - 1) << 16)) // max(width - 1, 1)
        fy = (np.arange(height, dtype=np.int64) * ((gh - 1) << 16)) // max(height - 1, 1)
        x0 = np.minimum(fx >> 16, gw - 2)
        y0 = np.minimum(fy >> 16, gh - 2)
        ax = (fx - (x0 << 16))[None, :, None]
        ay = (fy - (y0 << 16))[:, None, None]
        n00 = nodes[y0][:, x0]
        n01 = nodes[y0][:, x0 + 1]
        n10 = nodes[y0 + 1][:, x0]
        n11 = nodes[y0 + 1][:, x0 + 1]
        top = n00 * (65536 - ax) + n01 * ax
        bot = n10 * (65536 - ax) + n11 * ax
        img = (top * (65536 - ay) + bot * ay + (1 << 31)) >> 32
        # compress the range a little so the noise rarely clips
        img = 24 + (img * 208) // 256
        if noise:
            ctr = (np.arange(width * height * 3, dtype=np.uint64) + (np.uint64(seed) + np.uint64(7)) * np.uint64(0x100000001B3))
            h = _hash_u32(ctr)
            s = ((h & np.uint32(255)).astype(np.int64) + ((h >> np.uint32(8)) & np.uint32(255)).astype(np.int64) +
                 ((h >> np.uint32(16)) & np.uint32(255)).astype(np.int64) + (h >> np.uint32(24)).astype(np.int64))
            img = img + ((s - 510) * 43 // 1024).reshape(height, width, 3)  # sigma ~ 147.8 * 43/1024 ~ 6.2
    return np.clip(img, 0, 255).astype(np.uint8)
