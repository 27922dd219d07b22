"""``LoopClosureDetection`` with the reference's interface (loop_closure_detection/loop_closure_detection.py:16-83)
on the MI355X: MobileNetV3-small features -> L2 normalisation -> exact cosine search, all on the device.

Differences from the reference by design: the per-frame feature never leaves HBM (the reference moves it to
numpy, normalises it on the host and hands it to faiss, :45-48), the index is ``clslam_hip.flat_index.FlatIPIndex``
(attribute name ``faiss_index`` kept: ``ntotal`` / ``reconstruct`` / ``search`` behave like the faiss index the
reference builds at :35-36), and only the <= 100 candidate scores + positions of a search come back to the host,
where the threshold / id-gap / best-N filter of :58-76 runs on them.  ``display_matches`` is plotting."""
from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from clslam_hip import _lib
from clslam_hip.flat_index import FlatIPIndex, normalize_L2
from loop_closure_detection.config import LoopClosureDetection as Config
from loop_closure_detection.encoder import FeatureEncoder

SEARCH_K = 100    # loop_closure_detection.py:56


class LoopClosureDetection:
    def __init__(self, config: Config):
        self.threshold = config.detection_threshold
        self.id_threshold = config.id_threshold
        self.num_matches = config.num_matches

        lib = _lib.get_lib()                      # raises without libclslam_hip.so: there is no CPU path
        if lib.is_device and not torch.cuda.is_available():
            raise RuntimeError('libclslam_hip.so needs an MI355X (no GPU visible); there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if lib.is_device else torch.device('cpu')

        self.model = FeatureEncoder(self.device)
        self.faiss_index = FlatIPIndex(self.model.num_features, self.device)
        self.image_id_to_index = {}
        self.index_to_image_id = {}

    def _features(self, image: Tensor) -> Tensor:
        if len(image.shape) == 3:
            image = image.unsqueeze(dim=0)
        f = self.model(image).reshape(-1, self.model.num_features).contiguous()
        normalize_L2(f)                           # then the inner product is the cosine similarity (:47)
        return f

    def add(self, image_id: int, image: Tensor) -> None:
        f = self._features(image)
        if f.shape[0] != 1:
            raise ValueError('add() takes one image')    # the reference's squeeze()/expand_dims(…, 0) pair, :45-46
        self.faiss_index.add(f)
        self.image_id_to_index[image_id] = self.faiss_index.ntotal - 1
        self.index_to_image_id[self.faiss_index.ntotal - 1] = image_id

    def search(self, image_id: int) -> Tuple[List[int], np.ndarray]:
        index_id = self.image_id_to_index[image_id]
        distances, indices = self.faiss_index.search(self.faiss_index.row(index_id), SEARCH_K)
        distances, indices = distances[0], indices[0]
        # placeholders past ntotal, the query itself, weak matches, temporal neighbours (:58-70) -- one mask
        keep = (indices != -1) & (indices != index_id) & (distances > self.threshold) & \
               (np.abs(indices - index_id) > self.id_threshold)
        distances, indices = distances[keep][:self.num_matches], indices[keep][:self.num_matches]
        # like :75-76 the ids come back sorted while `distances` stays in similarity order
        image_ids = sorted(self.index_to_image_id[int(i)] for i in indices)
        return image_ids, distances

    def predict(self, image_0: Tensor, image_1: Tensor) -> float:
        """cosine similarity of two images' features (:78-83)"""
        f0, f1 = self._features(image_0), self._features(image_1)
        probe = FlatIPIndex(self.model.num_features, self.device)
        probe.add(f0[:1])
        return float(probe.scores(f1[:1])[0, 0].cpu())

    @staticmethod
    def display_matches(image_0, image_1, image_id_0, image_id_1, transformation, cosine_similarity):
        import matplotlib.pyplot as plt
        from slam.transform import string_tmat  # the caller's package (reference); imported late like :89-90
        imgs = [im.squeeze().cpu().detach().permute(1, 2, 0) if isinstance(im, Tensor) else im for im in (image_0, image_1)]
        filename = Path(f'./figures/sequence_00/matches/{image_id_0:04}_{image_id_1:04}.png')
        filename.parent.mkdir(parents=True, exist_ok=True)
        fig = plt.figure()
        for pos, img, title in ((211, imgs[0], image_id_0), (212, imgs[1], image_id_1)):
            plt.subplot(pos)
            plt.imshow(img)
            plt.axis('off')
            plt.title(title)
        plt.suptitle(f'cos_sim = {cosine_similarity:.4f} \n {string_tmat(transformation)}')
        plt.savefig(filename)
        plt.close(fig)
