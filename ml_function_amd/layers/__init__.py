from ..capture import capture_step
from .base import Layer
from .behavior_layer import MultHeadAttentionLayer, ProductAttentionLayer
from .core_layer import (AlignLayer, Dense, DnnLayer, HiddenLayer, IntraViewPoolingLayer, MergeScoreLayer, ResActivateLayer, ScoreLayer,
                         StackLayer)
from .interactive_layer import (CIN, AttentionBaseLayer, CrossLayer, ExtractLayer, FmLayer, InnerLayer, IPnnLayer, LinearLayer,
                                OPnnLayer, SparseEmbed)

__all__ = ["capture_step", "Layer", "InnerLayer", "FmLayer", "CrossLayer", "CIN", "SparseEmbed", "ProductAttentionLayer",
           "MultHeadAttentionLayer", "StackLayer", "ScoreLayer", "MergeScoreLayer", "HiddenLayer", "ResActivateLayer",
           "DnnLayer", "Dense", "IPnnLayer", "OPnnLayer", "LinearLayer", "AttentionBaseLayer", "ExtractLayer",
           "IntraViewPoolingLayer", "AlignLayer"]
