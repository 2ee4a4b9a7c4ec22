"""parse_config -- Python-module configs -> SimpleNamespace (counterpart of reference spectre_vit/configs/parser.py:5-27).

Same observable behaviour, written for Python >= 3.10 (the reference's ``SimpleNamespace(mapping)`` call needs 3.13):
* public module attributes plus the literal key ``__base__`` are collected (a ``_base_`` spelling is dropped with every
  other underscore name);
* with ``__base__ = "<file>.py"`` the sibling base module is merged ON TOP of the child (``mod |= base_mod``: base wins);
* the result is a SimpleNamespace.
``config_path`` is the slash path the scripts pass (``"spectre_vit/configs/spectre_vit_cifar100.py"``).
"""
import importlib
from types import SimpleNamespace


def module_to_dict(module):
    return {k: getattr(module, k) for k in dir(module) if not k.startswith("_") or k == "__base__"}


def parse_config(config_path: str) -> SimpleNamespace:
    dotted = config_path.replace("/", ".")
    if dotted.endswith(".py"):
        dotted = dotted[:-3]
    cfg = module_to_dict(importlib.import_module(dotted))
    if "__base__" in cfg:
        base = cfg["__base__"]
        base_dotted = ".".join(dotted.split(".")[:-1] + [base[:-3] if base.endswith(".py") else base])
        cfg.update(module_to_dict(importlib.import_module(base_dotted)))  # base overrides child, as in the reference
    return SimpleNamespace(**cfg)
