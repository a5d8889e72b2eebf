"""Reference checkpoint wire format (SURVEY next row f4): ``{'epoch', 'it', 'model_state', 'optimizer_state', 'version'}``
saved with torch.save to ``<name>.pth`` (tools/train_utils/train_utils.py:140-174).  ``model_state`` has the reference's
parameter names and layouts (sparse-conv weights in spconv-2.x (Cout, 3, 3, Cin)), ``optimizer_state`` is the
torch.optim.Adam state_dict the reference's OptimWrapper would write (gdmae_hip.optim.FlatAdamOneCycle.state_dict).
Loading goes through Detector3DTemplate.load_params_from_file / load_params_with_optimizer (reference
detector3d_template.py:361-442)."""
from __future__ import annotations

import torch

VERSION = "gdmae_hip+r01"


def model_state_to_cpu(model_state):
    out = type(model_state)()
    for k, v in model_state.items():
        out[k] = v.cpu()
    return out


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    return {"epoch": epoch, "it": it, "model_state": None if model is None else model_state_to_cpu(model.state_dict()),
            "optimizer_state": None if optimizer is None else optimizer.state_dict(), "version": VERSION}


def save_checkpoint(state, filename="checkpoint"):
    torch.save(state, f"{filename}.pth")
