"""Checkpoint hand-over with the reference's key names.

The reference moves the encode path's weights between its stages through two kinds of plain `torch.save` dictionaries:

* the stage-1 tokenizer checkpoint, read by `initialize_vision_modules` (src/model/setokim_arch.py:94-99):

      weights = torch.load(pretrain_vision_tokenizer, map_location='cpu')
      get_w   = lambda weights, keyword: {k.split(keyword + '.')[1]: v for k, v in weights.items() if keyword in k}
      self.vision_tower.load_state_dict(get_w(weights, 'tokenizer'), strict=False)

  i.e. every key that CONTAINS `tokenizer` is kept and cut after the first `tokenizer.` — the stage-1 model (`SeTok`, setok/model.py) holds the
  tokenizer as `self.tokenizer`, so its keys read `tokenizer.inner_encoder.layers.0.0.attn.qkv.weight`, ...  (`detokenizer.*` keys also contain
  the keyword and are cut the same way; `strict=False` drops the ones that do not exist in the tokenizer, and one that does exist under the same
  name overwrites the tokenizer's — a quirk of the reference that `select_by_keyword` reproduces and `test_host_cpu.py` pins.)

* the adapter checkpoint `mm_projector.bin`, written by `SetokimTrainer._save_checkpoint` (src/train/setokim_trainer.py:234-251: every named
  parameter whose name contains `mm_in_projector` / `mm_out_projector`) and read back by `initialize_vision_modules` (setokim_arch.py:115-120)
  through the same `get_w(weights, 'mm_in_projector')`.

This module writes and reads exactly those dictionaries, so a checkpoint written here loads in the reference and vice versa, plus the
optimiser state of `HeadTrainer` for resuming a run (the reference leaves that to the HF Trainer).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Mapping, Optional, Union

import torch

TOKENIZER_KEYWORD = "tokenizer"
ADAPTER_KEYWORDS = ("mm_in_projector", "mm_out_projector")          # setokim_trainer.py:243
ADAPTER_FILE = "mm_projector.bin"                                    # setokim_trainer.py:251

PathOrDict = Union[str, os.PathLike, Mapping[str, torch.Tensor]]


def _read(src: PathOrDict) -> Mapping[str, torch.Tensor]:
    if isinstance(src, Mapping):
        return src
    return torch.load(os.fspath(src), map_location="cpu")


def select_by_keyword(weights: Mapping[str, torch.Tensor], keyword: str) -> Dict[str, torch.Tensor]:
    """`get_w` of setokim_arch.py:95-96 / :117-118, to the letter: keys containing `keyword`, cut after the first `keyword + '.'`.
    A key that contains the keyword but not `keyword + '.'` raises IndexError there; it does here too."""
    return {k.split(keyword + ".")[1]: v for k, v in weights.items() if keyword in k}


# ---------------------------------------------------------------------------------------------------------------------------------------
# stage-1 tokenizer checkpoint
# ---------------------------------------------------------------------------------------------------------------------------------------
def tokenizer_checkpoint(tok: torch.nn.Module, include_tower: bool = False, prefix: str = TOKENIZER_KEYWORD + ".") -> Dict[str, torch.Tensor]:
    """The tokenizer's state under the stage-1 model's key names (`tokenizer.<name>`), on the CPU.  The frozen tower
    (`image_feature_encoder.*`) is left out unless asked for: the reference re-reads it from the CLIP checkpoint in `load_model`."""
    out = {}
    for k, v in tok.state_dict().items():
        if not include_tower and k.startswith("image_feature_encoder."):
            continue
        out[prefix + k] = v.detach().to("cpu").clone()
    return out


def save_tokenizer_checkpoint(tok: torch.nn.Module, path: Union[str, os.PathLike], include_tower: bool = False) -> Dict[str, torch.Tensor]:
    sd = tokenizer_checkpoint(tok, include_tower)
    torch.save(sd, os.fspath(path))
    return sd


def load_pretrained_tokenizer(tok: torch.nn.Module, src: PathOrDict):
    """setokim_arch.py:94-99.  Returns torch's (missing_keys, unexpected_keys) so a caller can see what `strict=False` let through."""
    return tok.load_state_dict(select_by_keyword(_read(src), TOKENIZER_KEYWORD), strict=False)


# ---------------------------------------------------------------------------------------------------------------------------------------
# adapter checkpoint (mm_projector.bin)
# ---------------------------------------------------------------------------------------------------------------------------------------
def adapter_checkpoint(named_params: Iterable, keys_to_match: Iterable[str] = ADAPTER_KEYWORDS) -> Dict[str, torch.Tensor]:
    """`get_mm_adapter_state_maybe_zero_3` (setokim_trainer.py:35-38) without the ZeRO-3 gather: the named parameters whose names contain
    one of `keys_to_match`, moved to the CPU, names untouched (`model.mm_in_projector.0.weight`, ...)."""
    keys = tuple(keys_to_match)
    return {k: t.detach().to("cpu").clone() for k, t in named_params if any(m in k for m in keys)}


def save_adapter_checkpoint(model: torch.nn.Module, output_dir: Union[str, os.PathLike], keys_to_match: Iterable[str] = ADAPTER_KEYWORDS) -> str:
    os.makedirs(os.fspath(output_dir), exist_ok=True)
    path = os.path.join(os.fspath(output_dir), ADAPTER_FILE)
    torch.save(adapter_checkpoint(model.named_parameters(), keys_to_match), path)
    return path


def load_pretrained_projector(projector: torch.nn.Module, src: PathOrDict, keyword: str = "mm_in_projector"):
    """setokim_arch.py:115-120."""
    return projector.load_state_dict(select_by_keyword(_read(src), keyword), strict=False)


# ---------------------------------------------------------------------------------------------------------------------------------------
# trainer state (resume)
# ---------------------------------------------------------------------------------------------------------------------------------------
def trainer_state(trainer) -> Dict[str, object]:
    """Everything `HeadTrainer` needs to continue bit-for-bit: step count, hyper-parameters, fp32 master weights and both AdamW moments."""
    cpu = lambda d: {n: t.detach().to("cpu").clone() for n, t in d.items()}
    return {"t": trainer.t, "lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd,
            "master": cpu(trainer.master), "exp_avg": cpu(trainer.m), "exp_avg_sq": cpu(trainer.v)}


def load_trainer_state(trainer, state: Mapping[str, object], load_hyper: bool = True) -> None:
    names = set(trainer.master)
    for part in ("master", "exp_avg", "exp_avg_sq"):
        got = set(state[part])
        if got != names:
            raise KeyError(f"trainer state '{part}' does not match the head's parameters: missing {sorted(names - got)[:4]}, "
                           f"unexpected {sorted(got - names)[:4]}")
    trainer.t = int(state["t"])
    if load_hyper:
        trainer.lr, trainer.betas, trainer.eps, trainer.wd = float(state["lr"]), tuple(state["betas"]), float(state["eps"]), float(state["weight_decay"])
    with torch.no_grad():
        for n in trainer.master:
            trainer.master[n].copy_(state["master"][n])
            trainer.m[n].copy_(state["exp_avg"][n])
            trainer.v[n].copy_(state["exp_avg_sq"][n])
            trainer.params[n].data.copy_(trainer.master[n])           # the low-precision weights are the rounded masters
    for enc in (getattr(trainer.tok, "inner_encoder", None), getattr(trainer.tok, "inter_encoder", None)):
        if enc is not None and hasattr(enc, "_packed"):
            enc._packed = {}


def save_training_checkpoint(trainer, output_dir: Union[str, os.PathLike]) -> Dict[str, str]:
    """One directory per checkpoint: `tokenizer.bin` (stage-1 key names, loadable by the reference's `pretrain_vision_tokenizer`) and
    `trainer_state.bin` (resume)."""
    d = os.fspath(output_dir)
    os.makedirs(d, exist_ok=True)
    paths = {"tokenizer": os.path.join(d, "tokenizer.bin"), "trainer": os.path.join(d, "trainer_state.bin")}
    save_tokenizer_checkpoint(trainer.tok, paths["tokenizer"])
    torch.save(trainer_state(trainer), paths["trainer"])
    return paths


def load_training_checkpoint(trainer, output_dir: Union[str, os.PathLike]) -> None:
    d = os.fspath(output_dir)
    load_pretrained_tokenizer(trainer.tok, os.path.join(d, "tokenizer.bin"))
    load_trainer_state(trainer, torch.load(os.path.join(d, "trainer_state.bin"), map_location="cpu"))
