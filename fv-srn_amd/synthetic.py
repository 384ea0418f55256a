"""Seeded synthetic scene-representation networks in the reference's shape conventions (no checkpoints exist offline):
what bench.py times and what the parity tests render.  Host-side numpy only."""
import numpy as np

from . import volnet_io


def random_arrays(*, C=32, layers=4, output_mode="density:direct", grid=None, fourier_std=None, seed=0, grid_scale=0.3, time_grids=1,
                  no_fourier=False, has_time=False, weight_gain=1.0):
    """The fp32 arrays of a seeded random SRN in the reference's shape conventions: nn.Linear default init U(+-1/sqrt(in)),
    NeRF block-identity Fourier matrix (network.py:55-63) unless fourier_std is given; grid = (channels, res).
    weight_gain scales every weight matrix: the default init shrinks the signal by ~1/6 (ReLU) per layer, so behind the 10 .. 22 layers of the
    reference's deep study networks (eval_NetworkConfigsGrid.py:37) the output would be the last bias; ~2.3 (ReLU) / ~2.0 (SnakeAlt) keep the
    dependence on the position alive without turning the stack chaotic in its own fp16 roundings.
    Returns dict(B, weights, biases, grids): what export_to_pyrenderer hands over, and what the CPU baseline of bench.py times."""
    rng = np.random.RandomState(seed)
    F = 0 if no_fourier else (C - 4) // 2  # (has_time: the time takes the padding channel 3, network.py:123-169 "extra" input)
    if no_fourier:
        B = np.zeros((0, 3), np.float32)
    elif fourier_std is None:
        blocks = [(2.0 ** i) * np.eye(3) for i in range((F + 2) // 3)]
        B = (np.concatenate(blocks, axis=0)[:F] * 2 * np.pi).astype(np.float32)
    else:
        B = (rng.randn(F, 3) * fourier_std * 2 * np.pi).astype(np.float32)
    G = grid[0] if grid else 0
    cout = 6 if output_mode.startswith("densitycurvature") else (
        4 if output_mode.startswith("rgbo") or output_mode.startswith("densitygrad") else 1)
    dims = [3 + int(has_time) + 2 * F + G] + [C] * (layers - 1) + [cout]
    weights, biases = [], []
    for i in range(layers):
        k = 1.0 / np.sqrt(dims[i])
        weights.append((rng.uniform(-k, k, (dims[i + 1], dims[i])) * weight_gain).astype(np.float32))
        biases.append(rng.uniform(-k, k, dims[i + 1]).astype(np.float32))
    if output_mode.startswith("rgbo"):  # reference network.py:404-405: positive rgba bias "to see something"
        biases[-1] = np.abs(biases[-1]) + 1.0
    tg = None
    if grid:
        tg = [(rng.randn(G, grid[1], grid[1], grid[1]) * grid_scale).astype(np.float32) for _ in range(time_grids)]
    return dict(B=B, weights=weights, biases=biases, grids=tg)


def random_network(*, C=32, layers=4, activation="SnakeAlt", param=1.0, output_mode="density:direct", grid=None,
                   fourier_std=None, seed=0, box_min=(0.0, 0.0, 0.0), box_size=(1.0, 1.0, 1.0), encoding=volnet_io.ENC_FLOAT,
                   grid_scale=0.3, time_grids=1, no_fourier=False, has_time=False, weight_gain=1.0):
    """random_arrays() as a VolnetData (fp16 weights in the stored layouts of the .volnet format)."""
    a = random_arrays(C=C, layers=layers, output_mode=output_mode, grid=grid, fourier_std=fourier_std, seed=seed, grid_scale=grid_scale,
                      time_grids=time_grids, no_fourier=no_fourier, has_time=has_time, weight_gain=weight_gain)
    return volnet_io.build_volnet(fourier_B=a["B"], weights=a["weights"], biases=a["biases"], activation=activation, activation_param=param,
                                  output_mode=output_mode, box_min=box_min, box_size=box_size, time_grids=a["grids"],
                                  grid_encoding=encoding, has_time=has_time)
