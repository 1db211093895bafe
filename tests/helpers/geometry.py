"""What "apply the EXIF orientation" means for an array of pixels (tests/test_gpu_geometry.py uses it for the expected values;
tests/test_exif_turns.py pins it)."""
import numpy as np


def upright(a, orientation):
    """EXIF orientation -> numpy (a is H x W x C or H x W)"""
    return {1: lambda x: x, 2: lambda x: x[:, ::-1], 3: lambda x: x[::-1, ::-1], 4: lambda x: x[::-1],
            5: lambda x: x.swapaxes(0, 1), 6: lambda x: np.rot90(x, -1), 7: lambda x: np.rot90(x, 2).swapaxes(0, 1)[...],
            8: lambda x: np.rot90(x, 1)}[orientation](a)
