"""Device-side data path in front of the hot path: the reference's CLIPFeatureExtractor image / label
transforms (evaluate.py:58-61, data/collator.py:82-91) and the ADE20K collate (data/collator.py:168-180)."""
from .preprocess import ADE20KCollator, ClipImagePreprocessor, ClipLabelPreprocessor

__all__ = ["ADE20KCollator", "ClipImagePreprocessor", "ClipLabelPreprocessor"]
