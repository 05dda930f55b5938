"""Layer shapes of the detector and the physique net used by tools/bench_conv.py and tools/pmc_shapes_report.py."""
SHAPES = [  # Hi, Wi, Cin, Cout, R, stride, pad
    (64, 64, 64, 64, 1, 1, 0), (64, 64, 64, 64, 3, 1, 1), (64, 64, 64, 256, 1, 1, 0),
    (64, 64, 256, 64, 1, 1, 0), (64, 64, 256, 128, 1, 1, 0), (64, 64, 128, 128, 3, 2, 1),
    (32, 32, 128, 512, 1, 1, 0), (32, 32, 512, 128, 1, 1, 0), (32, 32, 128, 128, 3, 1, 1),
    (32, 32, 256, 256, 3, 2, 1), (16, 16, 256, 1024, 1, 1, 0), (16, 16, 1024, 256, 1, 1, 0),
    (16, 16, 256, 256, 3, 1, 1), (16, 16, 512, 512, 3, 2, 1), (8, 8, 512, 2048, 1, 1, 0),
    (8, 8, 2048, 512, 1, 1, 0), (8, 8, 512, 512, 3, 1, 1), (64, 64, 256, 1152, 1, 1, 0),
    (128, 128, 64, 64, 3, 1, 1), (256, 256, 32, 32, 3, 1, 1), (256, 256, 64, 32, 3, 1, 1),
    (256, 256, 3, 64, 7, 2, 3),                          # stem
]


def images_for(n, shp):
    """images per launch: the tensors stay inside the 2 GiB range of 32-bit buffer offsets"""
    hi, wi, ci, co = shp[:4]
    if n * hi * wi * max(ci, co) * 4 >= 2**31 - 2**24:
        return max(1, int((2**31 - 2**24) // (hi * wi * max(ci, co) * 4)))
    return n
