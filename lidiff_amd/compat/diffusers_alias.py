"""``from diffusers import DPMSolverMultistepScheduler`` (tools/diff_completion_pipeline.py:6, models/models.py:17)
served by lidiff_amd.schedulers (the restatement of diffusers==0.18.0 for LiDiff's configuration)."""
from ..schedulers import DPMSolverMultistepScheduler  # noqa: F401

__version__ = "0.18.0+lidiff_amd"
