"""Base config (reference configs/default.py:1-2)."""
from spectre_vit.configs._presets import DEFAULT as _D

globals().update(_D)
