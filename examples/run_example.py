"""The build's counterpart of the reference's api/run_example.py (BASELINE configs[0], plumbing): the same three
calls -- Tester(model_weight_path, batch_size=64, workers=8, quiet=True); tester.test(example_video); print -- on an
MI355X.  The reference's inputs cannot be fetched offline (OpenFace binary, examples/utterance_1.mp4, the Google-Drive
checkpoint, the third-party ResNet50 weights), so this script first materialises stand-ins IN THE REFERENCE'S ON-DISK
FORMATS under a scratch directory:
    <dir>/examples/utterance_1_opface/utterance_1_aligned/frame_det_00_%06d.bmp   (309 synthetic aligned faces)
    <dir>/models/model_weights.pth.tar                  {'epoch', 'state_dict'}  (api/tester.py:47-49)
    <dir>/pytorch-benchmarks/ferplus/resnet50_ferplus_dag.pth                    (api/resnet50_extractor.py:35-36)
With real files in those places the same code runs unchanged.   usage: python examples/run_example.py [scratch_dir]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import synthetic, weights  # noqa: E402
from mimamo_net_amd.tester import Tester  # noqa: E402


def materialise(root, n_frames=309):
    from PIL import Image
    aligned = os.path.join(root, "examples", "utterance_1_opface", "utterance_1_aligned")
    os.makedirs(aligned, exist_ok=True)
    for i, frame in enumerate(synthetic.make_clip_u8(7, n_frames)):
        Image.fromarray(frame, "RGB").save(os.path.join(aligned, "frame_det_00_%06d.bmp" % (i + 1)))
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.make_two_stream_state_dict(seed=0).items()}
    torch.save({"epoch": 0, "state_dict": sd}, os.path.join(root, "models", "model_weights.pth.tar"))
    os.makedirs(os.path.join(root, "pytorch-benchmarks", "ferplus"), exist_ok=True)
    rs = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.make_resnet50_state_dict(seed=0).items()}
    torch.save(rs, os.path.join(root, "pytorch-benchmarks", "ferplus", "resnet50_ferplus_dag.pth"))


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="mimamo_example_")
    materialise(root)
    os.chdir(root)
    # ---- from here on: api/run_example.py, line for line in meaning
    example_video = 'examples/utterance_1.mp4'
    model_weight_path = 'models/model_weights.pth.tar'
    tester = Tester(model_weight_path, batch_size=64, workers=8, quiet=True)
    tester.test(example_video)  # warm-up (kernel load, workspaces) -- the reference's timing includes it
    torch.cuda.synchronize()
    tic = time.time()
    results = tester.test(example_video)
    torch.cuda.synchronize()
    took = time.time() - tic
    n_frames = results[list(results.keys())[0]].shape[0]
    print("Prediction takes {:.4f} seconds for {} frames, average {:.4f} seconds for one frame.".format(
        took, n_frames, took / n_frames))
    for video in results.keys():
        print("{} predictions".format(video))
        print(results[video])


if __name__ == "__main__":
    main()
