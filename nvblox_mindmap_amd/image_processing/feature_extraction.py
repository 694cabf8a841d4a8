"""Feature extractors of the mapping path (mirror of mindmap/image_processing/feature_extraction.py:33-620).

The CONTRACT is in scope, the networks are not: same class / method names, argument meaning, assertions and results as the
reference for everything around the DNN --

    FeatureExtractor.compute(rgb [b,h,w,3])                      :170-196   preprocess -> model -> resize -> NHWC -> zero pad
    FeatureExtractor.preprocess_image(rgb, (mean, std))          :222-254   u8 -> /255, (x - mean) / std, NCHW, bilinear resize to
                                                                            feature_image_size * model_downscale_factor (or the
                                                                            model's native input size)
    FeatureExtractor.train_dataset_mean_and_std()                :164-168   the per-extractor normalisation hook (identity by default,
                                                                            ImageNet statistics for DINOv2 :549-553)
    scale_image / _get_zero_padded_features / num_excess_features / model_downscale_factor      :110-129, 198-220, 270-277
    RgbFeatureExtractor                                          :556-590   the weight-free extractor (features = the resized image)
    get_feature_extractor / get_nvblox_feature_dim               :47-107

The reference's RADIO / DINOv2 / CLIP classes fetch their weights from the network (``torch.hub`` / ``clip.load``); here the
same classes take the network as an argument (``model=``): any module with the reference model's call convention.  Without one
they raise -- there is no silent stand-in.

Extension used by ``nvblox_integrate``: ``compute_lowres(rgb)`` hands over the model's own low-resolution output
``[h, w, C]`` and the size ``compute`` would resize it to; the native integration samples it in the kernel (bit-identical to
integrating ``compute(rgb)`` cast to float16, nvblox_mapping_helpers.py:256, without the [Hf, Wf, 768] image ever existing).
"""
import math
from abc import ABC, abstractmethod
from enum import Enum
from typing import Optional, Tuple

import torch

from ..nvblox_torch.constants import constants


class FeatureExtractorType(Enum):
    CLIP_RESNET50_FPN = "clip_resnet50_fpn"
    RADIO_V25_B = "radio_v25_b"
    DINO_V2_VITS14 = "dino_v2_vits14"
    RGB = "rgb"


def assert_square_and_batched_image(image):
    assert image.ndim == 4, "Expected BxHxWxC"
    assert image.shape[1] == image.shape[2], "Need square images"


def scale_image(tensor: torch.Tensor, target_size: Tuple[int, int], mode: str = "bilinear"):
    """[B,C,H,W] -> [B,C,*target_size] (bilinear, align_corners=False; :110-129)."""
    assert tensor.ndim == 4
    return torch.nn.functional.interpolate(tensor, size=target_size, mode=mode, align_corners=False)


def _held_output(extractor, rgb: torch.Tensor, hold: bool, run):
    """The one place an extractor reuses a network output: ``compute_lowres(rgb)`` followed by ``compute(rgb)`` on the SAME tensor
    object (``nvblox_integrate``'s fallback when the low-res hand-over does not apply) runs the network once.  The hand-over is
    explicit and short-lived: ``compute_lowres`` keeps (the image tensor itself, its version, the output); ``compute`` takes it
    only for that very object at that version, and takes it ONCE; ``release_lowres`` drops it.  Keeping the tensor alive is what
    makes the identity test sound -- an address + version key is not: a fresh per-frame tensor starts at version 0 again and
    the allocator hands the freed address out again, so the next frame would be served the previous frame's features (round-3
    advisor finding).  Not detectable: a write through a raw pointer into the held tensor between the two calls."""
    held = getattr(extractor, "_held", None)
    extractor._held = None
    if held is not None and held[0] is rgb and held[1] == rgb._version:
        out = held[2]
    else:
        out = run()
    if hold:
        extractor._held = (rgb, rgb._version, out)
    return out


class FeatureExtractor(ABC):
    """Abstract base of the extractors (:132-295)."""

    def __init__(self, feature_image_size: Optional[Tuple[int, int]] = None, pad_to_nvblox_dim: bool = False,
                 desired_output_size: Optional[Tuple[int, int]] = None, device: str = "cuda"):
        self.feature_image_size = feature_image_size
        self.pad_to_nvblox_dim = pad_to_nvblox_dim
        self.desired_output_size = desired_output_size
        self.device = device  # (the reference hard-codes "cuda"; a CPU device lets the contract be checked without a GPU)
        self.model = self.load()
        if self.model is not None:
            self.model.to(device).eval()
            for p in self.model.parameters():
                p.requires_grad = False
        assert self.embedding_dim() <= constants.feature_array_num_elements(), (
            f"Embedding dim: {self.embedding_dim()} is greater than nvblox's max feature size: "
            f"{constants.feature_array_num_elements()}. Rebuild nvblox with a larger feature size.")

    def train_dataset_mean_and_std(self):
        """Mean / std of the extractor's training set; identity here, overridden where a model wants normalised input (:164-168)."""
        return torch.tensor([0.0, 0.0, 0.0]), torch.tensor([1.0, 1.0, 1.0])

    def _features_bchw(self, rgb: torch.Tensor, hold: bool = False) -> torch.Tensor:
        assert rgb.ndim == 4
        assert rgb.shape[3] == 3
        return _held_output(self, rgb, hold, lambda: self._extract_features_impl(self.preprocess_image(rgb, self.train_dataset_mean_and_std())))

    def release_lowres(self) -> None:
        """Drop the model output ``compute_lowres`` keeps for a following ``compute`` of the same image (``nvblox_integrate`` calls
        this when it is done with a frame)."""
        self._held = None

    def compute(self, rgb: torch.Tensor):
        """rgb (b,h,w,3) -> features (b,H,W,F), float32 like the reference (:170-196)."""
        features_bchw = self._features_bchw(rgb)
        if self.desired_output_size is not None:
            features_bchw = scale_image(features_bchw, self.desired_output_size)
        features_bhwc = features_bchw.permute(0, 2, 3, 1)
        if self.pad_to_nvblox_dim:
            features_bhwc = self._get_zero_padded_features(features_bhwc)
        return features_bhwc

    @torch.no_grad()
    def compute_lowres(self, rgb: torch.Tensor):
        """Extension (see the module docstring): (model output [h, w, C8] float32 with C padded to a multiple of 8 by zero
        channels, the size ``compute`` resizes to) -- or (None, size) when ``compute`` would not produce a padded, resized
        single image the native path can reproduce."""
        size = self.desired_output_size
        if size is None or not self.pad_to_nvblox_dim or rgb.shape[0] != 1:
            return None, size
        low = self._features_bchw(rgb, hold=True)[0].to(torch.float32)  # [C, h, w]
        c8 = (low.shape[0] + 7) // 8 * 8
        if c8 != low.shape[0]:
            low = torch.cat([low, torch.zeros((c8 - low.shape[0],) + tuple(low.shape[1:]), device=low.device)], dim=0)
        return low.permute(1, 2, 0).contiguous(), size

    def _get_zero_padded_features(self, features_bhwc: torch.Tensor):
        assert_square_and_batched_image(features_bhwc)
        assert features_bhwc.shape[3] == self.embedding_dim(), \
            f"Features have incorrect embedding dimension: {features_bhwc.shape[3]} != {self.embedding_dim()}"
        n_batches, side = features_bhwc.shape[0], features_bhwc.shape[1]
        zeros = torch.zeros(n_batches, side, side, self.num_excess_features()).to(features_bhwc.device)
        return torch.cat((features_bhwc, zeros), dim=3)

    def num_excess_features(self):
        num_excess = constants.feature_array_num_elements() - self.embedding_dim()
        assert num_excess >= 0, (f"Embedding dim: {self.embedding_dim()} is less than nvblox's max feature size: "
                                 f"{constants.feature_array_num_elements()}. Rebuild nvblox with a larger feature size.")
        return num_excess

    def preprocess_image(self, rgb_bhwc: torch.Tensor, mean_and_std: Tuple[torch.Tensor, torch.Tensor]):
        """u8 -> [0,1] float (or the range assertion), normalise, NCHW, resize to the model's input (:222-254)."""
        mean, std = mean_and_std
        if rgb_bhwc.dtype == torch.uint8:
            rgb_bhwc = rgb_bhwc.float() / 255.0
        else:
            assert torch.max(rgb_bhwc) <= 1.0 and torch.min(rgb_bhwc) >= 0.0, "Image should be normalized to [0, 1]"
        rgb_bhwc = (rgb_bhwc - mean.to(device=rgb_bhwc.device)) / std.to(device=rgb_bhwc.device)
        rgb_bchw = rgb_bhwc.permute(0, 3, 1, 2)
        if self.feature_image_size is not None:
            required_input_size = (self.feature_image_size[0] * self.model_downscale_factor(),
                                   self.feature_image_size[1] * self.model_downscale_factor())
        else:
            required_input_size = self.model_input_size()
        assert required_input_size[0] % self.model_input_size()[0] == 0
        assert required_input_size[1] % self.model_input_size()[1] == 0
        return scale_image(rgb_bchw, required_input_size)

    @abstractmethod
    def embedding_dim(self):
        """Number of active elements in a feature vector"""

    @abstractmethod
    def model_input_size(self) -> Tuple[int, int]:
        """Native input size of the network (integer multiples work too)"""

    @abstractmethod
    def model_output_size(self) -> Tuple[int, int]:
        """Output size of the network at its native input size"""

    def model_downscale_factor(self) -> int:
        input_size, output_size = self.model_input_size(), self.model_output_size()
        assert input_size[0] % output_size[0] == 0
        assert input_size[1] % output_size[1] == 0
        assert input_size[0] / output_size[0] == input_size[1] / output_size[1]
        return int(input_size[0] / output_size[0])

    @abstractmethod
    def _extract_features_impl(self, rgb: torch.Tensor):
        """(b,3,h,w) preprocessed image -> (b,C,h',w') features"""

    @abstractmethod
    def load_model(self):
        """Return the network (or None)"""

    def load(self):
        """(The reference lets rank 0 fetch the weights first, :284-295; nothing is fetched here.)"""
        return self.load_model()


class RgbFeatureExtractor(FeatureExtractor):
    """Features = the resized RGB image (:556-590)."""

    @staticmethod
    def embedding_dim():
        return 3

    def model_input_size(self):
        return (32, 32)

    def model_output_size(self):
        return (32, 32)

    @staticmethod
    def load_model():
        return None

    @torch.no_grad()
    def _extract_features_impl(self, rgb_bchw: torch.Tensor):
        return rgb_bchw


class _InjectedModelExtractor(FeatureExtractor):
    """An extractor whose network is handed in (the reference downloads it)."""

    _what = "the network"

    def __init__(self, feature_image_size=None, pad_to_nvblox_dim=False, desired_output_size=None, model=None, device: str = "cuda"):
        if model is None:
            raise RuntimeError(f"{type(self).__name__} needs model=: {self._what} (the reference fetches it over the network; "
                               "this package does not, and does not substitute anything for it)")
        self._model_arg = model
        super().__init__(feature_image_size=feature_image_size, pad_to_nvblox_dim=pad_to_nvblox_dim,
                         desired_output_size=desired_output_size, device=device)

    def load_model(self):
        return self._model_arg


class RadioFeatureExtractorBase(_InjectedModelExtractor):
    """RADIO family (:298-337): ``model(x) -> (summary, features [b, h*w, C])``, 256 -> 16."""

    _what = "a RADIO model, called as model(rgb_bchw) -> (summary, features[b, tokens, C])"

    @torch.no_grad()
    def _extract_features_impl(self, rgb_bchw: torch.Tensor):
        _, features = self.model(rgb_bchw)
        output_size = int(math.sqrt(features.shape[1]))
        return features.view(rgb_bchw.shape[0], output_size, output_size, -1).permute(0, 3, 1, 2)

    def model_input_size(self):
        return (256, 256)

    def model_output_size(self):
        return (16, 16)


class RadioV25BFeatureExtractor(RadioFeatureExtractorBase):
    @staticmethod
    def embedding_dim():
        return 768


class DinoV2Vits14FeatureExtractor(_InjectedModelExtractor):
    """DINOv2 ViT-S/14 (:505-553): ``model.get_intermediate_layers(x, n=1)[0]`` [b, tokens, 384], 224 -> 16, ImageNet statistics."""

    _what = "a DINOv2 model with get_intermediate_layers(x, n=1) -> ([b, tokens, 384],)"

    @staticmethod
    def embedding_dim():
        return 384

    def model_input_size(self):
        return (224, 224)

    def model_output_size(self):
        return (16, 16)

    @torch.no_grad()
    def _extract_features_impl(self, rgb_bchw: torch.Tensor):
        features = self.model.get_intermediate_layers(rgb_bchw, n=1)[0]  # last layer's patch tokens
        output_size = int(math.sqrt(features.shape[1]))
        return features.view(rgb_bchw.shape[0], output_size, output_size, -1).permute(0, 3, 1, 2)

    def train_dataset_mean_and_std(self):
        return torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])


class ClipResNet50FpnFeatureExtractor(_InjectedModelExtractor):
    """CLIP ResNet-50 + feature pyramid (:373-468): 120 channels from pyramid level "res3", 256 -> 16, WebImageText statistics.
    ``model``: ``(backbone, pyramid_network)`` -- the CLIP visual trunk returning its five stages as a dict {"res1" .. "res5"} and the
    FPN over them (the reference builds both from ``clip.load("RN50")`` and torchvision) -- or one module doing both,
    ``model(rgb_bchw) -> {"res3": [b, 120, h, w], ...}``.  ``fpn_path``: a state dict for the pyramid network; as in the reference, a
    loaded FPN is frozen and one without a checkpoint stays trainable (:425-443)."""

    _what = "(CLIP RN50 trunk returning {'res1'..'res5'}, feature pyramid network) or one module returning {'res3': [b,120,h,w]}"

    def __init__(self, feature_image_size=None, pad_to_nvblox_dim=False, desired_output_size=None, fpn_path: Optional[str] = None,
                 model=None, device: str = "cuda"):
        pair = isinstance(model, (tuple, list))
        self.backbone, self.pyramid_network = (model[0], model[1]) if pair else (None, None)
        super().__init__(feature_image_size, pad_to_nvblox_dim, desired_output_size, model=torch.nn.ModuleList(model) if pair else model,
                         device=device)
        if pair:  # FeatureExtractor.__init__ froze everything: the pyramid is trainable unless its checkpoint was given (:431-441)
            if fpn_path is not None:
                self.pyramid_network.load_state_dict(torch.load(fpn_path, map_location=device, weights_only=True))
            for p in self.pyramid_network.parameters():
                p.requires_grad = fpn_path is None

    @staticmethod
    def embedding_dim():
        return 120

    def model_input_size(self):
        return (256, 256)

    def model_output_size(self):
        return (16, 16)

    def train_dataset_mean_and_std(self):
        return torch.tensor([0.48145466, 0.4578275, 0.40821073]), torch.tensor([0.26862954, 0.26130258, 0.27577711])

    @torch.no_grad()
    def _extract_features_impl(self, rgb_bchw: torch.Tensor):
        encoded = self.pyramid_network(self.backbone(rgb_bchw)) if self.backbone is not None else self.model(rgb_bchw)
        return encoded["res3"]


def get_nvblox_feature_dim(feature_extractor_type: FeatureExtractorType):
    dims = {FeatureExtractorType.CLIP_RESNET50_FPN: ClipResNet50FpnFeatureExtractor.embedding_dim(), FeatureExtractorType.RADIO_V25_B: RadioV25BFeatureExtractor.embedding_dim(),
            FeatureExtractorType.DINO_V2_VITS14: DinoV2Vits14FeatureExtractor.embedding_dim(),
            FeatureExtractorType.RGB: RgbFeatureExtractor.embedding_dim()}
    if feature_extractor_type not in dims:
        raise ValueError(f"Invalid feature extractor type: {feature_extractor_type}")
    return dims[feature_extractor_type]


def get_feature_extractor(feature_extractor_type: FeatureExtractorType, feature_image_size: Optional[Tuple[int, int]] = None,
                          desired_output_size: Optional[Tuple[int, int]] = None, pad_to_nvblox_dim: bool = False,
                          fpn_path: Optional[str] = None, model=None, device: str = "cuda"):
    """(:47-107) -- ``model``: the network of the RADIO / DINOv2 extractors (see the module docstring)."""
    kw = dict(feature_image_size=feature_image_size, desired_output_size=desired_output_size, pad_to_nvblox_dim=pad_to_nvblox_dim,
              device=device)
    if feature_extractor_type == FeatureExtractorType.RADIO_V25_B:
        return RadioV25BFeatureExtractor(model=model, **kw)
    if feature_extractor_type == FeatureExtractorType.DINO_V2_VITS14:
        return DinoV2Vits14FeatureExtractor(model=model, **kw)
    if feature_extractor_type == FeatureExtractorType.RGB:
        return RgbFeatureExtractor(**kw)
    if feature_extractor_type == FeatureExtractorType.CLIP_RESNET50_FPN:
        return ClipResNet50FpnFeatureExtractor(fpn_path=fpn_path, model=model, **kw)
    raise ValueError(f"Invalid feature extractor type: {feature_extractor_type}")
