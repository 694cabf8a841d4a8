#!/usr/bin/env python3
"""Generates tests/golden/feature_extraction.npz by IMPORTING the reference's mindmap/image_processing/feature_extraction.py
(authoring container only; the fixture -- seeds, shapes and the reference's OUTPUTS -- is committed, the reference's code is not).

The module's third-party imports that are absent here and never reached by what is exercised get import-only stubs (``clip``,
``torchvision``); ``nvblox_torch`` is this package under its drop-in alias (``install_as_nvblox_torch()``: only
``constants.feature_array_num_elements()`` is used).  The reference moves its normalisation constants to "cuda"
(feature_extraction.py:237); there is no GPU here, so ``Tensor.to(device="cuda")`` is redirected to the CPU for the duration of
the run -- arithmetic unchanged.

Cases (inputs are rebuilt by the test from numpy PCG64 seeds):
  rgb_*        RgbFeatureExtractor (:556-590) -- the weight-free extractor: u8 and float inputs, feature_image_size sizing,
               resize to desired_output_size, zero padding to the nvblox width (8 here)
  tiny_*       a subclass of the reference's FeatureExtractor written here around a seeded two-layer conv net (stride 4 -> a
               "model" with input 32x32 / output 8x8) with ImageNet statistics from train_dataset_mean_and_std: pins the hook,
               the ``feature_image_size x model_downscale_factor`` input sizing (:240-251) and the output chain (:186-195)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def seeded(shape, seed, lo=None, hi=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    if lo is not None:
        return rng.uniform(lo, hi, size=shape).astype(np.float32)
    return rng.standard_normal(shape).astype(np.float32)


def tiny_state():
    return {"0.weight": seeded((6, 3, 4, 4), 101) * 0.2, "0.bias": seeded((6,), 102) * 0.1,
            "2.weight": seeded((16, 6, 1, 1), 103) * 0.3, "2.bias": seeded((16,), 104) * 0.1}


def tiny_net():
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 4, stride=4), torch.nn.Tanh(), torch.nn.Conv2d(6, 16, 1))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in tiny_state().items()})
    return net


def main():
    assert os.path.isdir("/root/reference"), "the reference is only available in the authoring container"
    import nvblox_mindmap_amd

    nvblox_mindmap_amd.install_as_nvblox_torch()
    from nvblox_torch.constants import constants

    stub("clip", load=None)
    stub("clip.model", ModifiedResNet=object)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.ops = stub("torchvision.ops", FeaturePyramidNetwork=object)
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        if k.get("device") == "cuda":
            k = dict(k, device="cpu")
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_cpu
    import mindmap.image_processing.feature_extraction as RF

    out = {}
    try:
        constants.set_feature_array_num_elements(8)
        rgb_u8 = (seeded((1, 48, 64, 3), 11, 0.0, 1.0) * 255.0).astype(np.uint8)
        rgb_f = seeded((2, 40, 40, 3), 12, 0.0, 1.0)
        ex = RF.RgbFeatureExtractor(feature_image_size=None, pad_to_nvblox_dim=True, desired_output_size=(24, 24))
        out["rgb_u8_pad_24"] = ex.compute(torch.from_numpy(rgb_u8)).numpy()
        ex = RF.RgbFeatureExtractor(feature_image_size=(32, 32), pad_to_nvblox_dim=False, desired_output_size=None)
        out["rgb_f_fis32"] = ex.compute(torch.from_numpy(rgb_f)).numpy()
        ex = RF.RgbFeatureExtractor(feature_image_size=(32, 32), pad_to_nvblox_dim=True, desired_output_size=(30, 30))
        out["rgb_f_fis32_pad_30"] = ex.compute(torch.from_numpy(rgb_f)).numpy()
        out["rgb_num_excess"] = np.array(ex.num_excess_features())

        constants.set_feature_array_num_elements(24)

        class Tiny(RF.FeatureExtractor):
            @staticmethod
            def embedding_dim():
                return 16

            def model_input_size(self):
                return (32, 32)

            def model_output_size(self):
                return (8, 8)

            @staticmethod
            def load_model():
                return None  # (the base class would call .cuda() on a model; the net is attached below)

            def _extract_features_impl(self, rgb_bchw):
                with torch.no_grad():
                    return self.net(rgb_bchw)

            def train_dataset_mean_and_std(self):
                return torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])

        for name, kw, img in (("tiny_native_pad_20", dict(feature_image_size=None, pad_to_nvblox_dim=True, desired_output_size=(20, 20)), rgb_u8),
                              ("tiny_fis16_pad_36", dict(feature_image_size=(16, 16), pad_to_nvblox_dim=True, desired_output_size=(36, 36)), rgb_u8),
                              ("tiny_fis8_raw", dict(feature_image_size=(8, 8), pad_to_nvblox_dim=False, desired_output_size=None), rgb_f)):
            ex = Tiny(**kw)
            ex.net = tiny_net().eval()
            out[name] = ex.compute(torch.from_numpy(img)).numpy()
            out[name + "_model_input"] = np.array(ex.preprocess_image(torch.from_numpy(img), ex.train_dataset_mean_and_std()).shape)
        out["tiny_downscale"] = np.array(ex.model_downscale_factor())
    finally:
        torch.Tensor.to = orig_to
        constants.set_feature_array_num_elements(768)
    path = os.path.join(HERE, "feature_extraction.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
