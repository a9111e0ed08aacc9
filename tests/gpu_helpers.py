"""Build iqlpref_amd objects from golden-file contents (GPU tests)."""
import numpy as np
import torch

import iqlpref_amd as ia

DEV = "cuda:0"


def make_nets(hyper, nets, device=DEV):
    qf_p, vf_p, actor_p = nets
    S, A, H = hyper["s_dim"], hyper["a_dim"], hyper["hidden"]
    E, NH = hyper.get("n_critics", 2), hyper.get("n_hidden", 2)
    q = ia.TwinQ(S, A, hidden_dim=H, n_hidden=NH) if E == 2 else \
        ia.EnsembleQ(S, A, hidden_dim=H, n_hidden=NH, n_critics=E)
    v = ia.ValueFunction(S, hidden_dim=H, n_hidden=NH)
    cls = ia.DeterministicPolicy if hyper["deterministic"] else ia.GaussianPolicy
    actor = cls(S, A, 1.0, hidden_dim=H, n_hidden=NH, dropout=hyper["dropout"])
    for mod, params in ((q, qf_p), (v, vf_p), (actor, actor_p)):
        sd = {k: torch.from_numpy(np.asarray(a)) for k, a in params.items()}
        missing = mod.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return q.to(device), v.to(device), actor.to(device)


def make_trainer(hyper, nets, mode, device=DEV, **kw):
    q, v, actor = make_nets(hyper, nets, device)
    vo = torch.optim.Adam(v.parameters(), lr=3e-4)
    qo = torch.optim.Adam(q.parameters(), lr=3e-4)
    ao = torch.optim.Adam(actor.parameters(), lr=3e-4)
    tr = ia.ImplicitQLearning(
        max_action=1.0, actor=actor, actor_optimizer=ao, q_network=q, q_optimizer=qo,
        v_network=v, v_optimizer=vo, iql_tau=hyper["iql_tau"], beta=hyper["beta"],
        max_steps=hyper["max_steps"], discount=hyper["discount"], tau=hyper["tau"],
        device=device, precision=mode, **kw)
    return tr


def make_buffer(hyper, data, device=DEV):
    buf = ia.ReplayBuffer(hyper["s_dim"], hyper["a_dim"], hyper["n_rows"] + 7, device)
    buf.load_d4rl_dataset({k: np.asarray(v) for k, v in data.items()})
    return buf


def module_params(mod):
    return {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
