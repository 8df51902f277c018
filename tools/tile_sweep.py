"""Per-shape block-tile sweep of the conv engine on the ResNet 1x1 shapes (mm_conv2d_nhwc force_tile):
1 = 128x128, 2 = 128x64, 4 = 256x64, 5 = 128x256 (8 waves)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb
# (name, B, H, W, Cin, Cout, k, stride, pad, residual)
shapes = [("64->256 res", 2048, 56, 56, 64, 256, 1, 1, 0, 1), ("256->64", 2048, 56, 56, 256, 64, 1, 1, 0, 0),
          ("128->512 res", 2048, 28, 28, 128, 512, 1, 1, 0, 1), ("512->128", 2048, 28, 28, 512, 128, 1, 1, 0, 0),
          ("256->1024 res", 2048, 14, 14, 256, 1024, 1, 1, 0, 1), ("1024->256", 2048, 14, 14, 1024, 256, 1, 1, 0, 0),
          ("512->2048 res", 2048, 7, 7, 512, 2048, 1, 1, 0, 1), ("2048->512", 2048, 7, 7, 2048, 512, 1, 1, 0, 0),
          ("4096^3", 1, 64, 64, 4096, 4096, 1, 1, 0, 0)]
tiles = [int(t) for t in sys.argv[1:]] or [1, 2, 4, 5]
for name, B, H, W, Ci, Co, k, st, pad, res in shapes:
    for tile in tiles:
        if tile in (5, 6) and Co % 256: continue
        if tile == 7 and Co % 128: continue
        try:
            print(name, end=": ")
            cb.run(B, H, W, Ci, Co, k, st, pad, tile, 10, res, 0)
        except AssertionError as e:
            print("tile", tile, "refused", e)
