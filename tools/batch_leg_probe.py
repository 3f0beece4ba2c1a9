"""Developer probe (round 5): why does the batch leg of the default bench line read lower than the standalone cfg4 line?
One process, config 4 (8 pairs of 1920 x 1080 as one lock-step group per step, four lanes), 64 steps, median of three regions:
  A  plain planes, fresh process          B  lane 0 writing into a torch tensor (the gather buffer), like the leg
  C  plain planes after a config-3 job ran and was closed in the same process (what the leg's position in the line is)
usage (GPU box): python tools/batch_leg_probe.py"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402

flow2d = importlib.import_module("cuda-flow2d_amd")
batch = importlib.import_module("cuda-flow2d_amd.batch")
import torch  # noqa: E402

torch.cuda.set_device(0)
batch.init(backend="nccl", device=torch.device("cuda", 0))
args = argparse.Namespace(pipeline=4, max_lanes=4, step_group=0, batch_mode="groups", no_graph=False, algorithm=0, steps=64, warmup=3)
cfg4 = bench.WORKLOADS[bench.BATCH_WORKLOAD]


def rate(tag, out_tensor=None):
    job = bench.Job(flow2d, batch, bench.BATCH_WORKLOAD, cfg4, args, 0, 0, 1, out_tensor=out_tensor)
    t = float(np.median(bench.timed_region(job, batch, torch, 64, 3, 3)))
    job.close()
    print("%-60s %8.1f pairs/s  %.3f ms per step" % (tag, 64 * 8 / t, t / 64 * 1e3), flush=True)


rate("A  plain planes, fresh process")
pitch_floats = flow2d.hip_lib().flow2d_plane_pitch_bytes(cfg4["w"]) // 4
local = torch.zeros((2, 8, cfg4["h"], pitch_floats), dtype=torch.float32, device="cuda:0")
rate("B  lane 0 writes into a torch tensor", local)
rate("A' plain planes again")
cfg3 = bench.WORKLOADS["cfg3_4096_gradient"]
job = bench.Job(flow2d, batch, "cfg3_4096_gradient", cfg3, args, 0, 0, 1)
bench.timed_region(job, batch, torch, 20, 3, 1)
job.close()
rate("C  plain planes after a closed config-3 job")
rate("C' torch tensor after a closed config-3 job", local)

# ---- which of the sampling legs leaves the process slower? (round 5: the batch leg read 8-9 % low after them) --------------------
import types  # noqa: E402

first_pair = bench.workload_pair("cfg3_4096_gradient", cfg3, 0)
params = flow2d.OpticalFlow.params(cfg3["levels"], cfg3["scale"], cfg3["outer"], cfg3["inner"], cfg3["alpha"], 0.001, 0.001,
                                   cfg3["median"], cfg3["sigma"], 0)
sample = types.SimpleNamespace(flow2d=flow2d, cfg=cfg3, local_rank=0, first_pair=first_pair, params=params, sync=lambda: None)
bench.measured_copy_peak(flow2d, 0)
rate("D  after measured_copy_peak (512 MiB device-to-device copy)")
bench.per_sweep_sample(sample)
rate("E  after per_sweep_sample (per-sweep launches, a context of its own)")
bench.roofline_sample(sample)
rate("F  after roofline_sample (eager passes with per-launch events, then a replayed lone pair)")
rate("F' again")
