"""The rotating point-cloud animation of the reference's visualiser (utils.py:78-179 there).

The point cloud comes from the GPU (pipeline.triangulate_point_cloud: one DLT launch per image
pair with the pixel cameras K[R|T]); drawing is matplotlib on the host, as in the reference.
What differs from the reference is only how the drawing is fed: the camera centre is the closed
form -M^-1 p4 of P = [M | p4] instead of a null-space SVD per camera, all cameras of the figure go
into one quiver call, and the number of frames and the output directory are arguments
(defaults: the reference's 120 frames of 3 degrees, output/rotation_dataset_<name>.gif).
"""
from __future__ import annotations

import os

import numpy as np

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover

    def tqdm(it=None, **kw):
        return it


# visualization.py:5-35 of the reference, one colour per image pair.  Names matplotlib does not know
# ("rose", "mustard") are kept: the reference fails on them too, from the 23rd pair on.
COLORS = [
    "blue", "green", "red", "cyan", "magenta", "yellow", "black", "orange", "purple", "brown", "pink", "gray", "olive", "navy",
    "lightblue", "lightgreen", "coral", "beige", "indigo", "lime", "maroon", "teal", "rose", "mustard", "turquoise", "sienna",
    "plum", "orchid", "lavender",
]

FRAME_STEP_DEGREES = 3  # utils.py:170: np.arange(0, 360, 3)
FRAME_INTERVAL_MS = 100
GIF_DPI = 60  # utils.py:176


def camera_centres_and_axes(P):
    """Centres (3 x n) and viewing directions (3 x n) of a list of 3x4 cameras.

    Reference: utils.py:92-102 takes the null vector of every P and divides by its last entry;
    for P = [M | p4] with M invertible that vector is (-M^-1 p4, 1).  The viewing direction is the
    third row of M (utils.py:100)."""
    P = np.asarray(P, np.float64).reshape(-1, 3, 4)
    centres = -np.linalg.solve(P[:, :, :3], P[:, :, 3:4])[:, :, 0]
    return centres.T.copy(), P[:, 2, :3].T.copy()


def plot_3d_points_and_cameras(X, P, ax, color):
    """Reference: utils.py:78-105.  X is 4xN homogeneous, P a list of 3x4 cameras."""
    X = np.asarray(X, np.float64)
    X3 = X / X[-1, :]
    ax.scatter(X3[0, :], X3[1, :], X3[2, :], s=1, color=color)
    c, v = camera_centres_and_axes(P)
    ax.quiver(c[0], c[1], c[2], v[0], v[1], v[2], color="r", linewidth=1.5)


def pixel_cameras(sfm_data_dict):
    """K[R_i | T_i] for every image (utils.py:141-149)."""
    K = np.asarray(sfm_data_dict["K"], np.float64)
    R = sfm_data_dict["absolute_rotations"]
    T = sfm_data_dict["refined_Ts"]
    return [K @ np.hstack((np.asarray(R[i], np.float64), np.reshape(T[i], (3, 1)))) for i in range(sfm_data_dict["nr_images"])]


def draw_scene(sfm_data_dict, clouds, colors=COLORS):
    """The static figure of utils.py:130-159: one coloured cloud and two red camera arrows per image pair."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fig = plt.figure(figsize=(10, 8))
    ax = fig.add_subplot(111, projection="3d")
    ax.set_xlabel("X")
    ax.set_ylabel("Y")
    ax.set_zlabel("Z")
    ax.set_title("3D Points and Camera Positions")
    ax.invert_yaxis()
    P = pixel_cameras(sfm_data_dict)
    for i, X in enumerate(clouds):
        plot_3d_points_and_cameras(X, [P[i], P[i + 1]], ax, colors[i])
    return fig, ax


def visualize_sfm_results_with_rotation(sfm_data_dict, dataset_name, colors=COLORS, clouds=None, frames=None, out_dir="output"):
    """Reference: utils.py:108-179.  Writes <out_dir>/rotation_dataset_<name>.gif and returns its path.

    `clouds` (one 4xN array per image pair) defaults to the GPU triangulation of the result
    dictionary; `frames` defaults to the reference's azimuths 0, 3, ..., 357."""
    import matplotlib.animation as animation
    import matplotlib.pyplot as plt

    if clouds is None:
        from .pipeline import triangulate_point_cloud

        clouds = triangulate_point_cloud(sfm_data_dict)
    fig, ax = draw_scene(sfm_data_dict, clouds, colors)
    angles = np.arange(0, 360, FRAME_STEP_DEGREES) if frames is None else np.asarray(frames)
    bar = tqdm(total=len(angles), desc="Generating Frames", unit="frame")

    def update_frame(angle):
        ax.view_init(elev=30, azim=float(angle), vertical_axis="y")
        if bar is not None:
            bar.update(1)

    anim = animation.FuncAnimation(fig, update_frame, frames=angles, interval=FRAME_INTERVAL_MS)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"rotation_dataset_{dataset_name}.gif")
    anim.save(path, dpi=GIF_DPI, writer="pillow")
    if bar is not None:
        bar.close()
    plt.close(fig)
    print(f"Animation saved to {path}")
    return path
