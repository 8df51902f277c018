"""ResNet50 trunk at different batch sizes: per-launch times of the conv engine (library measurement hook), to see
whether producer->consumer reuse through the 256 MB Infinity Cache shows up when the activations of a sub-batch fit."""
import ctypes, os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib, weights
from mimamo_net_amd.resnet50_extractor import Resnet50_Extractor

dev = torch.device("cuda:0")
L = _lib.lib()
rn = Resnet50_Extractor(state_dict=weights.make_resnet50_state_dict(seed=0), device=dev)
TOTAL = 1024
for bs in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 1024]:
    x = torch.rand(bs, 224, 224, 4, device=dev)
    for _ in range(2):
        rn.get_vec(x, channels_last4=True)
    torch.cuda.synchronize()
    os.environ["MM_PROF_DUMP"] = "/tmp/sweep_%d.csv" % bs
    ms = (ctypes.c_double * 5)(); work = (ctypes.c_double * 5)(); n = (ctypes.c_int64 * 5)()
    L.mm_profile_begin()
    rn.get_vec(x, channels_last4=True)
    L.mm_profile_end(ms, work, n)
    reps = TOTAL // bs
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        rn.get_vec(x, channels_last4=True)
    t1.record(); torch.cuda.synchronize()
    wall = t0.elapsed_time(t1) / reps
    print("bs %4d: conv %.3f ms (%.1f TF) transforms %.3f ms  | wall %.3f ms/batch = %.1f frames/s" %
          (bs, ms[0], work[0] / ms[0] / 1e9, ms[3], wall, bs / wall * 1e3), flush=True)
    agg = collections.OrderedDict()
    for r in csv.reader(open("/tmp/sweep_%d.csv" % bs)):
        if r[0] != "0":
            continue
        tag = ",".join(r[3:]).split(" t")[0]
        tag = " ".join(tag.split(" ")[1:])  # drop M=
        a = agg.setdefault(tag, [0.0, 0.0, 0])
        a[0] += float(r[2]); a[1] += float(r[1]); a[2] += 1
    for tag, (t, w, c) in list(agg.items())[:14]:
        print("     %-28s x%d  %.4f ms/frame-batch  %.1f TF" % (tag, c, t / c, w / t / 1e9))
