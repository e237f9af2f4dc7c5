"""Single-episode policy frequency through ``OtterInference``: the probe of the reference's
``tests/test_otter_interface.py:6-49`` (50 calls with black frames and a random pose, ``reset()``, 50 more calls timed,
``50 / seconds``), run against this build on the GPU.

    python scripts/policy_frequency.py [--clip-model ViT-L/14] [--calls 50]

The reference reads a trained checkpoint folder; there is none offline, so a random-init model of the default
configuration is written to a temporary folder in the reference's layout (``run.yaml`` + ``checkpoint_0.pt``) and
loaded through the constructor like a trained one.  Without ``OTTER_CLIP_BPE`` a synthetic merges table of the real
file's shape stands in for the BPE vocabulary (token ids differ from the real ones; the arithmetic per call does not).
Prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_synthetic_merges(path: str, n: int = 49152 - 256 - 2) -> None:
    """Header + n made-up merge pairs, so that start / end-of-text get ids 49406 / 49407 as with the real file."""
    from otter_amd.tokenizer import bytes_to_unicode
    base = list(bytes_to_unicode().values())
    syms = base + [b + "</w>" for b in base]
    pairs, seen, i = [], set(), 0
    while len(pairs) < n:
        a, b = syms[100 + (i // 400) % 400], syms[101 + i % 400]
        i += 1
        if (a, b) not in seen:
            seen.add((a, b))
            pairs.append((a, b))
    with open(path, "w", encoding="utf-8") as fh:
        fh.write("#version: synthetic\n")
        fh.write("\n".join(f"{a} {b}" for a, b in pairs) + "\n")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clip-model", default=None, help="CLIP backbone name (default: the configuration's, ViT-L/14)")
    ap.add_argument("--calls", type=int, default=50)
    args = ap.parse_args()

    import torch
    from PIL import Image
    from otter_amd.checkpoint import save_checkpoint
    from otter_amd.config import ExperimentConfig, ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.otter_interface import OtterInference
    from otter_amd.params import otter_specs, random_state_dict

    mc, sc = ModelConfig(), SharedConfig()
    if args.clip_model:
        mc.clip_model = args.clip_model
    with tempfile.TemporaryDirectory() as folder:
        if not os.environ.get("OTTER_CLIP_BPE"):
            merges = os.path.join(folder, "merges.txt")
            write_synthetic_merges(merges)
            os.environ["OTTER_CLIP_BPE"] = merges
        model = OTTER(mc, sc)
        model.load_state_dict(random_state_dict(otter_specs(mc, sc), seed=0), strict=True)
        save_checkpoint(model, ExperimentConfig(model_cfg=mc, shared_cfg=sc), folder, 0)
        del model
        policy = OtterInference(model_ckpt_folder=folder, ckpt_id=0)

        res = policy.args.shared_cfg.image_size
        images = {k: Image.new("RGB", (res, res)) for k in policy.model.camera_keys}
        rng = np.random.default_rng(0)

        def run(n: int) -> float:
            t0 = time.time()
            for _ in range(n):
                out = policy(images, "text", rng.random(6), rng.random(1))
                assert out.shape == (policy.model.action_horizon, policy.model.action_dim) and np.isfinite(out).all()
            return time.time() - t0

        run(args.calls)                  # the reference's first loop: fills the window (and warms everything up)
        policy.reset()
        seconds = run(args.calls)        # timed: PIL -> upload -> towers -> policy -> download -> absolute decode, per call
        print(json.dumps({"metric": "policy inference frequency (single episode, OtterInference.__call__)",
                          "value": args.calls / seconds, "unit": "Hz", "calls": args.calls, "ms_per_call": 1e3 * seconds / args.calls,
                          "clip_model": mc.clip_model, "device": torch.cuda.get_device_name(0)}))


if __name__ == "__main__":
    main()
