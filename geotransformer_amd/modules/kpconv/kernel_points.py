"""Kernel-point dispositions for KPConv (15 points, one fixed at the centre).

The coordinates are the optimised disposition the reference ships as
geotransformer/modules/kpconv/dispositions/k_015_center_3D.ply (fp64, unit radius; produced offline by
kernel_points.py:62-386 of the reference, which is out of scope here).  `load_kernels` reproduces the
run-time part of the reference's loader (kernel_points.py:389-455): cast to fp32, random rotation about z,
N(0, 0.01^2) noise, scaling by the convolution radius -- consuming numpy's global RNG in the same order, so
that a model built under `np.random.seed(s)` gets the same kernel points as the reference.
"""
import numpy as np

K015_CENTER_3D = np.array([
    [0.0, 0.0, 0.0],
    [-0.4982061244651975, 0.4182679671551277, 0.1173671831925196],
    [-0.24123564899318725, -0.3421404836489065, -0.5115480969173494],
    [-0.2828808007398322, -0.5861426591615737, 0.11553227719987667],
    [0.29054036421745955, -0.10093209154426704, -0.5850910017747533],
    [0.428200390449578, 0.39929883025634566, -0.3068181339821517],
    [-0.635864927863347, -0.08196440772765984, -0.16090402983021967],
    [-0.4318108191655937, -0.14729416644105348, 0.4783095747872299],
    [-0.0446660016247351, 0.2797321413633876, 0.5972330819203726],
    [0.22552417024986607, -0.344625435411299, 0.5079465901181037],
    [0.6388921231157457, -0.16914905918114065, -0.011901081494663384],
    [-0.22552414870165857, 0.3446254511305206, -0.5079465890257602],
    [0.4905466554531465, 0.26880703235056747, 0.35219206363799427],
    [0.25233083792271604, -0.5970665260439326, -0.12951142127628598],
    [0.03415393794300986, 0.6585834126638934, 0.04513958377027336],
], dtype=np.float64)


def load_kernels(radius, num_kpoints=15, dimension=3, fixed='center'):
    """(num_kpoints, 3) fp32 kernel points for one KPConv layer (reference kernel_points.py:389-455)."""
    if num_kpoints != 15 or dimension != 3 or fixed != 'center':
        raise ValueError('only the 15-point, 3-D, centre-fixed disposition is available '
                         f'(got {num_kpoints} points, {dimension}-D, fixed={fixed!r})')
    kernel_points = K015_CENTER_3D.astype(np.float32)
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)
    kernel_points = radius * kernel_points
    kernel_points = np.matmul(kernel_points, R)
    return kernel_points.astype(np.float32)
