"""Drop-in ``nn.Module`` replacements for the reference's hot-path modules."""
from .base import HipModule, ParamArena
from .clip import (ClipArch, ImageEncoderCLIP, ImageEncoderCLIPFull, TextEncoderCLIP, TextEncoderCLIPPooler,
                   TEXT_B, TEXT_L, VIT_B16, VIT_L14)
from .decoder import DecoderBlock, DecoderLayer, PromptDecoder, PromptLayer
from .hier import (CrossABlock, FTNBlock, FTNDecoder, HierarchicalCrossA, HierarchicalSelfA, SelfABlock,
                   SRTransformerCrossA, SRTransformerDecoder, SRTransformerSelfA)
from .loss import AuxiliaryLoss, ContrastiveLoss, CrossEntropyLoss, NPairLoss
from .model import BaseModelWithText, ContrastiveModel, TextToPatch
from .compose import DenseClip, PromptFTN
from .score import ScoreMapTail
from . import ftn  # model/ftn.py's Decoder / Transformer keep their (generic) names inside this submodule
from .swin import SWIN_B, SWIN_S, SWIN_T, SwinArch, SwinTransformer

__all__ = ["HipModule", "ParamArena", "ClipArch", "ImageEncoderCLIP", "ImageEncoderCLIPFull", "TextEncoderCLIP",
           "TextEncoderCLIPPooler", "DecoderBlock", "DecoderLayer", "PromptDecoder", "PromptLayer", "AuxiliaryLoss", "ContrastiveLoss", "CrossEntropyLoss", "NPairLoss",
           "BaseModelWithText", "ContrastiveModel", "TextToPatch", "PromptFTN", "DenseClip", "ScoreMapTail", "CrossABlock", "FTNBlock", "FTNDecoder", "HierarchicalCrossA",
           "HierarchicalSelfA", "SelfABlock", "SRTransformerCrossA", "SRTransformerDecoder", "SRTransformerSelfA", "VIT_B16", "VIT_L14", "TEXT_B", "TEXT_L", "SwinTransformer", "SwinArch",
           "SWIN_T", "SWIN_S", "SWIN_B", "ftn"]
