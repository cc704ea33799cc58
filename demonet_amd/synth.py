"""Build-owned, counter-based synthetic data generator.

Not `torch.rand`: values depend only on (seed, tensor key, element index), so the authoring
container, the GPU box and any torch version agree bit-for-bit on inputs and weights.
splitmix64 -> 24-bit uniforms; normals via Box-Muller in float64, cast to float32.
"""
import json
import os

import numpy as np

_CALIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib.json")
_calib_cache = None


CALIBRATED_SEEDS = (0, 1, 2)


def calibration(model_name: str, seed: int = 0):
    """Per-conv weight multipliers measured once at authoring time (tests/golden/calibrate_synth.py): they play the
    role training plays for real checkpoints -- keeping every layer's output near unit variance under the
    fixed synthetic BN statistics. Pure data; identical on every machine. Tables exist for the weight seeds
    in CALIBRATED_SEEDS; other seeds get uncalibrated weights (still valid, but activations drift)."""
    global _calib_cache
    if _calib_cache is None:
        _calib_cache = json.load(open(_CALIB_PATH)) if os.path.exists(_CALIB_PATH) else {}
    return _calib_cache.get(model_name, {}).get(str(seed), {})


_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _stream(seed: int, key: str, n: int, lane: int = 0) -> np.ndarray:
    base = np.uint64((fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15) ^ (lane * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base) & _MASK
    return _splitmix64(ctr)


def uniform(seed: int, key: str, n: int, lane: int = 0) -> np.ndarray:
    """float32 uniform in [0, 1) with 24 random bits."""
    return ((_stream(seed, key, n, lane) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def normal(seed: int, key: str, n: int) -> np.ndarray:
    u1 = ((_stream(seed, key, n, 1) >> np.uint64(11)).astype(np.float64) + 1.0) * (2.0 ** -53)   # (0,1]
    u2 = (_stream(seed, key, n, 2) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def images(seed: int, n: int, h: int, w: int) -> np.ndarray:
    """[n, 3, h, w] float32 uniform [0,1): synthetic stand-in for decoded images (engine.py:86 feeds such tensors)."""
    return uniform(seed, "images", n * 3 * h * w).reshape(n, 3, h, w)


def state_dict(graph, seed: int = 0, calibrated: bool = True):
    """Synthetic weights keyed by the reference's state_dict names (numpy arrays).

    Scales keep activations O(1) through the network (conv std = gain/sqrt(fan_in)) and give the class
    logits a realistic spread (std ~ graph builder's logit_std) so per-class candidate lists are tie-free.
    """
    out = {}
    calib = calibration(graph.name, seed) if calibrated else {}
    for p in graph.params:
        n = int(np.prod(p.shape)) if len(p.shape) else 1
        if p.kind == "conv_w":
            v = normal(seed, p.key, n) * np.float32(p.std * calib.get(p.key, 1.0))
        elif p.kind == "bias":
            v = normal(seed, p.key, n) * np.float32(p.std)
        elif p.kind == "bn_gamma":
            v = uniform(seed, p.key, n) + np.float32(0.5)
        elif p.kind == "bn_beta":
            v = normal(seed, p.key, n) * np.float32(0.1)
        elif p.kind == "bn_mean":
            v = normal(seed, p.key, n) * np.float32(0.1)
        elif p.kind == "bn_var":
            v = uniform(seed, p.key, n) + np.float32(0.5)
        elif p.kind == "bn_nbt":
            out[p.key] = np.zeros((), dtype=np.int64)
            continue
        elif p.kind == "scale20":
            v = np.full(n, 20.0, dtype=np.float32) * (uniform(seed, p.key, n) * np.float32(0.2) + np.float32(0.9))
        else:
            raise ValueError(p.kind)
        out[p.key] = v.reshape(p.shape).astype(np.float32)
    return out
