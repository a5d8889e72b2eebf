"""Seeded parameters of the BEV-backbone + CenterHead parity fixture (no imports beyond torch: used by the tests AND by
tests/golden/make_golden_head.py on the reference modules)."""
import torch


def seeded_head_state(net, seed):
    """Seeded parameters for the BEV-backbone + CenterHead parity fixture (tests/golden/make_golden_head.py uses this very
    function on the reference modules): state_dict order, N(0, 1/fan_in) weights, |N(0, 0.25)| + 0.5 BatchNorm scales,
    N(0, 0.25) biases; the heat-map output bias keeps its -2.19 initialisation."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in net.state_dict().items():
        if not v.dtype.is_floating_point or "running" in k:
            continue
        if ".hm." in k and k.endswith(".bias") and v.numel() <= 8:
            sd[k] = v.detach().clone().cpu()
            continue
        t = torch.randn(tuple(v.shape), generator=g) * (0.5 if v.dim() == 1 else 1.0 / max(1, v[0].numel()) ** 0.5)
        if v.dim() == 1 and k.endswith("weight"):
            t = t.abs() + 0.5
        sd[k] = t.float()
    return sd
