"""Loader of dataset5_geometry.npz (written by make_dataset5_geometry.py) for the GPU test and the benchmark: the pipeline's
inputs as its front-end would hand them over, and a stand-in for find_correspondences that returns the stored matches."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_fixture():
    with np.load(os.path.join(HERE, "dataset5_geometry.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}  # in memory: the stand-in matcher must not decompress the file on every call
    K = [[float(v) for v in row] for row in g["K"]]
    x_pairs = [g[f"x_pair_{k}"] for k in range(18)]
    desc = g["init_pair_desc"].astype(np.float32)
    init = {"init_pair_indices": [int(v) for v in g["init_pair"]], "x_init": [g["x_init_0"], g["x_init_1"]], "init_pair_desc": desc}
    paths = [f"image_{i}" for i in range(10)]
    n0 = desc.shape[1]
    Kinv = np.linalg.inv(np.asarray(K, np.float64))

    def matcher(tag):
        def find_correspondences(img_path, desc_X, X0, Kmat):
            """The stored matches of find_correspondences (src/utils.py:65-93) for this image: against all descriptors of the
            initial pair, or against those of the run's inliers (the caller is told apart by the size of desc_X)."""
            i = paths.index(img_path)
            full = np.asarray(desc_X).shape[1] == n0
            xy = g[f"corr_full_{i}_xy"] if full else g[f"{tag}corr_inl_{i}_xy"]
            ti = g[f"corr_full_{i}_ti"] if full else g[f"{tag}corr_inl_{i}_ti"]
            x = np.float32(xy).T.reshape(2, -1)
            return np.float32(np.asarray(X0)[:, ti]), Kinv @ np.vstack((x, np.ones((1, x.shape[1]))))

        return find_correspondences

    return g, K, x_pairs, init, paths, matcher
