"""bench.py's N > 1 code path on the one GPU of this box: `--dist` makes the world-size-1 run build the
`nccl` (= RCCL) process group and end every step with all_gather_into_tensor of the tile buffer + the resolve of the gathered
buffer - the calls an 8-GPU run makes, with one rank.  The resolved frame must be the single-GPU entry point's, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_nccl_group_of_one_rank(vpt, scene03, dev03, tmp_path):
    out = tmp_path / "frame.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist", "--resolution", "200", "--spp", "3", "--steps", "2", "--warmup", "1",
                        "--no-cold", "--no-others", "--cpu-sample", "0", "--dump-frame", str(out)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "nccl group of one rank" in line["config"]["parallelism"] and line["value"] > 0
    p = vpt.PathtraceParams(resolution=200, samples=9, shader="volpathtrace", bounces=64)
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 9)      # (1 warm-up + 2 timed steps) x 3 spp
    frame = np.load(out)
    assert frame.shape == (ref.height, ref.width, 4)
    assert np.array_equal(frame.view(np.uint32), vpt.get_render(ref).view(np.uint32))
