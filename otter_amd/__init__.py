"""MI355X-native implementation of OTTER's observation -> action-chunk hot path.

Host-side mirror of ``otter/policy`` of the reference; the compute runs in ``libotter_hip.so``.
"""
from .config import ExperimentConfig, ModelConfig, SharedConfig  # noqa: F401

__all__ = ["ModelConfig", "SharedConfig", "ExperimentConfig"]
