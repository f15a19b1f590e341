"""TEST INFRASTRUCTURE: torch-CPU restatement of ctu.models / ctu.trainers (see oracle/__init__.py)."""
from . import nets, model  # noqa: F401
