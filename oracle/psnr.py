"""Oracle (test infrastructure): the reference's parity metric.

Restates ``compute_psnr`` of python_coreml_stable_diffusion/torch2coreml.py:59-74 and the
gate ``ABSOLUTE_MIN_PSNR = 35`` (torch2coreml.py:77; tests/test_stable_diffusion.py:33).
"""
import numpy as np

ABSOLUTE_MIN_PSNR = 35.0   # torch2coreml.py:77 - the reference's hard floor
TARGET_PSNR_FP16 = 60.0    # our gate for fp16-HIP vs fp32-oracle


def compute_psnr(a, b):
    """20*log10((max|b| + 1e-5) / (rmse + 1e-10)); ``b`` is the trusted signal."""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    max_b = np.abs(b).max()
    sumdeltasq = np.sum((a - b) ** 2) / a.size
    return float(20.0 * np.log10((max_b + 1e-5) / (np.sqrt(sumdeltasq) + 1e-10)))
