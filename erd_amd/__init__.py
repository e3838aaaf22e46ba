"""erd_amd -- MI355X-native ERD incremental-detection training step.

Public surface mirrors the reference's plugin interface for this path:
``MODELS`` / ``TASK_UTILS`` registries, ``Config.fromfile`` and the registered classes
(``GFLIncrementERD``, ``GFL``, ``ResNet``, ``FPN``, ``GFLHead``, ``GFLHeadIncrementERD``, losses, task utils).
All compute goes through ``liberd_hip.so`` (``include/erd_hip.h``); there is no CPU fallback."""
from .config import Config, ConfigDict  # noqa: F401
from .registry import MODELS, TASK_UTILS, Registry  # noqa: F401
from .structures import DetDataSample, InstanceData  # noqa: F401
from . import modules  # noqa: F401  (registers the classes)
from .modules import parse_losses  # noqa: F401

__version__ = "0.1.0"
