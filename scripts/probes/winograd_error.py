#!/usr/bin/env python3
"""Numerical error of Winograd F(2x2, 3x3) against the direct 3x3 convolution for one 64 -> 64 layer on 5x5 boards, both in
f32 arithmetic (numpy float32: transforms, products and the K = 64 accumulation), judged against an fp64 direct convolution.
Companion of scripts/probes/mfma_probe.hip (probe_wino): the go / no-go of VERDICT round 3, item 4 asks for <= 1e-6 relative."""
import numpy as np

rng = np.random.default_rng(0)
B, C, N = 64, 64, 5
x = np.maximum(rng.standard_normal((B, C, N, N)), 0).astype(np.float32)       # post-ReLU activations
w = (rng.standard_normal((C, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)

def direct(x, w, dt):
    xp = np.zeros((B, C, N + 2, N + 2), dt); xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, C, N, N), dt)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("bcyx,oc->boyx", xp[:, :, dy:dy + N, dx:dx + N].astype(dt), w[:, :, dy, dx].astype(dt)).astype(dt)
    return out

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)

def winograd(x, w):
    U = np.einsum("ij,ocjk,lk->ocil", G, w, G).astype(np.float32)              # G g G^T
    xp = np.zeros((B, C, 8, 8), np.float32); xp[:, :, 1:1 + N, 1:1 + N] = x    # rows/cols -1 .. 6
    out = np.zeros((B, C, 6, 6), np.float32)
    for ty in range(3):
        for tx in range(3):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = np.einsum("ij,bcjk,lk->bcil", BT, d, BT).astype(np.float32)    # B^T d B
            M = np.einsum("bcil,ocil->boil", V, U).astype(np.float32)          # 16 GEMMs over the channels
            out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,bojk,lk->boil", AT, M, AT).astype(np.float32)
    return out[:, :, :N, :N]

ref = direct(x.astype(np.float64), w.astype(np.float64), np.float64)
scale = np.abs(ref).max()
for name, y in (("direct f32", direct(x, w, np.float32)), ("Winograd F(2x2,3x3) f32", winograd(x, w))):
    err = np.abs(y.astype(np.float64) - ref)
    print(f"{name:26s} max |err| / max|y| = {err.max() / scale:.2e}   rms relative = {np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")
