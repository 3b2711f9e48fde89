"""Deterministic synthetic radar frames.

No dataset travels with this repo, so every test, the bench and the golden-vector generator draw
their inputs from here.  A frame is what the reference's pre-processor hands to the graph
constructor (``preprocessor/radarscenes/dataset_creation.py:203-223``): ``X_cc`` [N,2],
``V_cc_compensated`` [N,2], ``rcs`` [N,1], ``timestamp`` [N,1] -- all float64 holding
float32-representable values (the datasets store float32).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class RadarFrame:
    X: np.ndarray          # [N,2] f64 spatial coordinates (car coordinates)
    V: np.ndarray          # [N,2] f64 ego-motion compensated velocity
    rcs: np.ndarray        # [N,1] f64
    timestamp: np.ndarray  # [N,1] f64

    @property
    def n(self) -> int:
        return self.X.shape[0]


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a.astype(np.float32).astype(np.float64))


def _clustered_frame(rng: np.random.Generator, n_clusters: int, pts_per_cluster: int, n_clutter: int,
                     x_range, y_range, n_timestamps: int) -> RadarFrame:
    xs, vs = [], []
    for _ in range(n_clusters):
        centre = np.array([rng.uniform(*x_range), rng.uniform(*y_range)])
        heading = rng.uniform(0.0, 2.0 * np.pi)
        rot = np.array([[np.cos(heading), -np.sin(heading)], [np.sin(heading), np.cos(heading)]])
        local = rng.normal(0.0, 1.0, size=(pts_per_cluster, 2)) * np.array([1.5, 0.7])
        xs.append(centre + local @ rot.T)
        v_obj = rng.normal(0.0, 5.0, size=(1, 2))
        vs.append(v_obj + rng.normal(0.0, 0.3, size=(pts_per_cluster, 2)))
    clutter_x = np.stack([rng.uniform(*x_range, size=n_clutter), rng.uniform(*y_range, size=n_clutter)], axis=1)
    clutter_v = rng.normal(0.0, 1.0, size=(n_clutter, 2))
    clutter_v[rng.uniform(size=n_clutter) < 0.6] = 0.0          # exactly-zero velocities: 90 deg branch
    xs.append(clutter_x)
    vs.append(clutter_v)
    X = np.concatenate(xs, axis=0)
    V = np.concatenate(vs, axis=0)
    n = X.shape[0]
    perm = rng.permutation(n)                                    # sensors do not deliver points cluster by cluster
    X, V = X[perm], V[perm]
    rcs = rng.normal(-5.0, 10.0, size=(n, 1))
    t_base = 1.0e6 + np.arange(n_timestamps, dtype=np.float64) * 17.0
    timestamp = t_base[rng.integers(0, n_timestamps, size=n)].reshape(n, 1)
    return RadarFrame(_f32(X), _f32(V), _f32(rcs), _f32(timestamp))


def radarscenes_frame(frame_idx: int = 0, n_clusters: int = 40, pts_per_cluster: int = 35,
                      n_clutter: int = 1600) -> RadarFrame:
    """RadarScenes-shaped frame, N = 40*35 + 1600 = 3000; FoV x in [0,100], y in [-50,50]
    (``configurations/configuration_radarscenes.yml:8``)."""
    rng = np.random.Generator(np.random.PCG64(1234 + frame_idx))
    return _clustered_frame(rng, n_clusters, pts_per_cluster, n_clutter, (0.0, 100.0), (-50.0, 50.0), 30)


def nuscenes_frame(frame_idx: int = 0) -> RadarFrame:
    """nuScenes-shaped sparse sweep accumulation, N = 8*15 + 180 = 300, 6 sweeps."""
    rng = np.random.Generator(np.random.PCG64(1234 + frame_idx))
    return _clustered_frame(rng, 8, 15, 180, (-100.0, 100.0), (-100.0, 100.0), 6)


def small_frame(n: int, seed: int = 0, zero_velocity_fraction: float = 0.3, duplicates: int = 0) -> RadarFrame:
    """Small hand-sized frames for the oracle-vs-reference fixtures; can carry exact duplicates."""
    rng = np.random.Generator(np.random.PCG64(99 + seed))
    X = rng.uniform(-5.0, 5.0, size=(n, 2))
    V = rng.normal(0.0, 2.0, size=(n, 2))
    V[rng.uniform(size=n) < zero_velocity_fraction] = 0.0
    for d in range(min(duplicates, n // 2)):
        X[n - 1 - d] = X[d]
    rcs = rng.normal(-5.0, 10.0, size=(n, 1))
    timestamp = rng.integers(0, 4, size=(n, 1)).astype(np.float64) * 10.0 + 100.0
    return RadarFrame(_f32(X), _f32(V), _f32(rcs), _f32(timestamp))


def stress_cloud(n_overlay: int = 33, extra_clutter: int = 1000) -> RadarFrame:
    """100 000-point cloud: 33 RadarScenes-shaped frames overlaid in one FoV + 1000 clutter points."""
    frames = [radarscenes_frame(10_000 + i) for i in range(n_overlay)]
    rng = np.random.Generator(np.random.PCG64(4321))
    cx = np.stack([rng.uniform(0, 100, size=extra_clutter), rng.uniform(-50, 50, size=extra_clutter)], axis=1)
    X = np.concatenate([f.X for f in frames] + [cx])
    V = np.concatenate([f.V for f in frames] + [np.zeros((extra_clutter, 2))])
    rcs = np.concatenate([f.rcs for f in frames] + [rng.normal(-5, 10, size=(extra_clutter, 1))])
    ts = np.concatenate([f.timestamp + 1000.0 * i for i, f in enumerate(frames)] +
                        [np.full((extra_clutter, 1), 5.0e5)])
    return RadarFrame(_f32(X), _f32(V), _f32(rcs), _f32(ts))


def concat_frames(frames):
    """Batch layout used on the device: rows of all frames back to back + ``frame_ptr`` [B+1] (int64)."""
    ptr = np.zeros(len(frames) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([f.n for f in frames])
    cat = lambda name: np.ascontiguousarray(np.concatenate([getattr(f, name) for f in frames], axis=0))
    return RadarFrame(cat("X"), cat("V"), cat("rcs"), cat("timestamp")), ptr
