"""Drop-in for utils/evaluation/RatingMetrics.py."""
import numpy as np


def RMSE(error, num):
    """RatingMetrics.py:4-6."""
    return np.sqrt(error / num)


def MAE(error_mae, num):
    """RatingMetrics.py:8-9."""
    return (error_mae / num)
