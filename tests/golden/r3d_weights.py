"""Weights of R3DNet((1, 1, 1, 1)) drawn from a numpy Generator in the reference's state_dict order (72 entries):
Conv3d kaiming-uniform-like scale, BatchNorm gamma ~ 1 + 0.1 n, beta ~ 0.1 n, running stats at their initial values.
Shared by make_goldens_r3d.py (which checks keys and shapes against the imported reference) and the tests."""
import math

import numpy as np


def r3d_weights(rng, layer_sizes=(1, 1, 1, 1), dtype=np.float32):
    sd = {}

    def conv(name, cout, cin, k):
        fan_in = cin * k[0] * k[1] * k[2]
        sd[name + ".temporal_spatial_conv.weight"] = (rng.standard_normal((cout, cin) + tuple(k)) * math.sqrt(2.0 / fan_in)).astype(dtype)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(dtype)
        sd[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(dtype)
        sd[name + ".running_mean"] = np.zeros(c, dtype)
        sd[name + ".running_var"] = np.ones(c, dtype)
        sd[name + ".num_batches_tracked"] = np.zeros((), np.int64)

    def block(pre, cin, cout, down):
        if down:                                   # registration order of SpatioTemporalResBlock.__init__
            conv(pre + ".downsampleconv", cout, cin, (1, 1, 1))
            bn(pre + ".downsamplebn", cout)
        conv(pre + ".conv1", cout, cin, (3, 3, 3))
        bn(pre + ".bn1", cout)
        conv(pre + ".conv2", cout, cout, (3, 3, 3))
        bn(pre + ".bn2", cout)

    conv("conv1", 64, 3, (3, 7, 7))
    bn("bn1", 64)
    cin = 64
    for li, (cout, n) in enumerate(zip((64, 128, 256, 512), layer_sizes)):
        name = f"conv{li + 2}"
        block(name + ".block1", cin, cout, li > 0)
        for b in range(n - 1):
            block(f"{name}.blocks.{b}", cout, cout, False)
        cin = cout
    return sd
