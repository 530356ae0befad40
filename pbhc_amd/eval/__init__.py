"""Offline evaluation metrics over recorded rollouts (metrics half): drop-in for the arithmetic of the reference's
humanoidverse/measure_traj.py, sample_eps.py and ratio_eps.py."""
from .metrics import (blend_motion, eval_accuracy, eval_batch_traj, eval_smoothness, first_termination_ratio, load_traj_data,  # noqa: F401
                      trajectory_tables)
