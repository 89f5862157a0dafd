"""Argument checks with the reference's exact messages (its tests regex-match them).

Mirrors /root/reference/mlx_audio_primitives/_validation.py:10-91.
"""

from __future__ import annotations


def validate_positive(value, name: str) -> None:
    if value <= 0:
        raise ValueError(f"{name} must be positive, got {value}")


def validate_non_negative(value, name: str) -> None:
    if value < 0:
        raise ValueError(f"{name} must be non-negative, got {value}")


def validate_range(value, name: str, min_val=None, max_val=None, min_inclusive=True,
                   max_inclusive=True) -> None:
    if min_val is not None:
        if min_inclusive and value < min_val:
            raise ValueError(f"{name} must be >= {min_val}, got {value}")
        elif not min_inclusive and value <= min_val:
            raise ValueError(f"{name} must be > {min_val}, got {value}")
    if max_val is not None:
        if max_inclusive and value > max_val:
            raise ValueError(f"{name} must be <= {max_val}, got {value}")
        elif not max_inclusive and value >= max_val:
            raise ValueError(f"{name} must be < {max_val}, got {value}")
