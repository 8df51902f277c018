import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
import conv_bench as cb
# (B,H,W,Cin,Cout,k,stride,pad, res)
shapes = [("64->256 res", 2048, 56, 56, 64, 256, 1, 1, 0, 1), ("256->64", 2048, 56, 56, 256, 64, 1, 1, 0, 0),
          ("128->512 res", 2048, 28, 28, 128, 512, 1, 1, 0, 1), ("512->128", 2048, 28, 28, 512, 128, 1, 1, 0, 0),
          ("256->1024 res", 2048, 14, 14, 256, 1024, 1, 1, 0, 1), ("1024->256", 2048, 14, 14, 1024, 256, 1, 1, 0, 0),
          ("512->2048 res", 2048, 7, 7, 512, 2048, 1, 1, 0, 1), ("2048->512", 2048, 7, 7, 2048, 512, 1, 1, 0, 0)]
for name, B, H, W, Ci, Co, k, st, pad, res in shapes:
    for tile in (1, 5, 4, 2):
        if tile == 5 and Co % 256: continue
        try:
            print(name, end=": ")
            cb.run(B, H, W, Ci, Co, k, st, pad, tile, 10, res, 0)
        except AssertionError as e:
            print("tile", tile, "refused", e)
