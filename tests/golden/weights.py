"""Test-side name for the package's deterministic synthetic weights / inputs (calm-vit-dte_amd/synthetic_weights.py,
loaded by path so that make_golden.py can use it without importing the package)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "calm-vit-dte_amd", "synthetic_weights.py")
_spec = importlib.util.spec_from_file_location("calm_synthetic_weights", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

make_tensor, make_params, make_input, NoiseStream, _rng = (_mod.make_tensor, _mod.make_params, _mod.make_input,
                                                           _mod.NoiseStream, _mod._rng)
