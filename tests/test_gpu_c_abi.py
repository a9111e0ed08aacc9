"""The boundary is a C-ABI library: tests/c_abi/abi_smoke.c (plain C, gcc, no Python, no torch)
drives libiqlhip.so; the same state built through the Python binding must give the same bits.
-m gpu."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def lcg_fill(n, seed, scale):
    out = np.empty(n, dtype=np.float32)
    s = np.uint32(seed)
    with np.errstate(over="ignore"):
        for i in range(n):
            s = np.uint32(s * np.uint32(1664525) + np.uint32(1013904223))
            out[i] = (np.float32(int(s) >> 8) * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * np.float32(scale)
    return out


@pytest.mark.parametrize("NH,H,kind", [(2, 64, 0), (3, 40, 1)])
def test_plain_c_program_matches_the_python_binding(tmp_path, NH, H, kind):
    """kind: 0 = the tuned step, 1 = the general layer-wise step (iqlhip_trainer_step_kind)."""
    import iqlpref_amd as ia
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no C compiler / ROCm headers on this box")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "iqlpref_amd")
    subprocess.run([cc, "-O1", "-std=c99", os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"),
                    "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    "-L", libdir, "-l:libiqlhip.so", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe, str(NH), str(H)], check=True, capture_output=True, text=True, timeout=120).stdout.split("\n")
    assert out[0] == f"total_it 12 step_kind {kind}"
    got_c = np.array([[float.fromhex(x) for x in line.split()] for line in out[1:13]], dtype=np.float32)

    S, A, B, N, STEPS = 11, 3, 32, 500, 12
    q, v = ia.TwinQ(S, A, hidden_dim=H, n_hidden=NH), ia.ValueFunction(S, hidden_dim=H, n_hidden=NH)
    actor = ia.GaussianPolicy(S, A, 1.0, hidden_dim=H, n_hidden=NH)
    mods = [q.q1, q.q2, v.v, actor.net]
    with torch.no_grad():
        for n, m in enumerate(mods):
            for li, lin in enumerate(m.linears()):
                w = lcg_fill(lin.weight.numel(), 1000 + (n * (NH + 1) + li) * 2, 0.25).reshape(lin.weight.shape)
                b = lcg_fill(lin.bias.numel(), 1000 + (n * (NH + 1) + li) * 2 + 1, 0.25)
                lin.weight.copy_(torch.from_numpy(w))
                lin.bias.copy_(torch.from_numpy(b))
    q, v, actor = q.to(DEV), v.to(DEV), actor.to(DEV)
    tr = ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=torch.optim.Adam(actor.parameters(), lr=3e-4),
        q_network=q, q_optimizer=torch.optim.Adam(q.parameters(), lr=3e-4),
        v_network=v, v_optimizer=torch.optim.Adam(v.parameters(), lr=3e-4),
        iql_tau=0.7, beta=3.0, max_steps=1000, discount=0.99, tau=0.005, device=DEV, precision="fp32", seed=2024)
    data = {"observations": lcg_fill(N * S, 1, 2.0).reshape(N, S), "actions": lcg_fill(N * A, 2, 2.0).reshape(N, A),
            "rewards": lcg_fill(N, 3, 1.0), "next_observations": lcg_fill(N * S, 4, 2.0).reshape(N, S),
            "terminals": (np.arange(N) % 37 == 0).astype(np.float32)}
    buf = ia.ReplayBuffer(S, A, N, DEV)
    buf.load_d4rl_dataset(data)
    got_py = tr.train_steps(buf, STEPS, B, graph_unroll=4).cpu().numpy()
    np.testing.assert_array_equal(got_c, got_py)
