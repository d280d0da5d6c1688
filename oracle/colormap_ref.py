"""numpy restatement of the reference's colour mapping.  TEST INFRASTRUCTURE ONLY.

data-to-pics/src/main.rs:139-144:
    let color = ui::GRADIENT.eval_continuous((ui::AMPLITUDE_SCALE * value).into());
    *pixel = Rgb([color.r, color.g, color.b]);
with ui::GRADIENT = colorous::INFERNO and AMPLITUDE_SCALE = 1.0 / 0.5 (ui/src/lib.rs:113-123).
colorous 1.0.16 (Cargo.lock:389-392) is not vendored; its sequential gradients follow d3-scale-chromatic's
`ramp`: n colours, index floor(t * n) clamped to [0, n - 1]; a NaN or negative t saturates to entry 0 in
the float -> usize cast.  "Parity unpinned": the reference holds no image fixtures.
"""
import numpy as np

AMPLITUDE_SCALE = np.float32(1.0) / np.float32(0.5)


def colormap(values: np.ndarray, palette: np.ndarray, scale=AMPLITUDE_SCALE) -> np.ndarray:
    """[rows, cols] float32 -> [rows, cols, 3] uint8 through ``palette`` ([n, 3] uint8)."""
    n = len(palette)
    t = (np.float32(scale) * values.astype(np.float32)).astype(np.float64)   # f32 multiply, then .into() f64
    with np.errstate(invalid="ignore"):
        x = np.floor(t * float(n))
        idx = np.where(x >= 0.0, np.minimum(x, n - 1), 0.0)                  # NaN and negatives -> 0
    return palette[idx.astype(np.int64)]
