"""alegnn_amd -- MI355X (gfx950) implementation of the GraphFilter / LSIGF hot path of alelab-upenn/graph-neural-networks.

Layout mirrors the reference package so that call sites read the same:
    alegnn.utils.graphML.GraphFilter            ->  alegnn_amd.utils.graphML.GraphFilter
    alegnn.utils.graphML.LSIGF                  ->  alegnn_amd.utils.graphML.LSIGF
    alegnn.utils.graphML.EdgeVariantGF          ->  alegnn_amd.utils.graphML.EdgeVariantGF
    alegnn.utils.graphML.NodeVariantGF / NVGF   ->  alegnn_amd.utils.graphML.NodeVariantGF / NVGF
    alegnn.utils.graphML.HiddenState / GatedGRNN->  alegnn_amd.utils.graphML.HiddenState / GatedGRNN
    alegnn.modules.architectures.SelectionGNN   ->  alegnn_amd.modules.architectures.SelectionGNN   (+ LocalGNN, NodeVariantGNN,
                                                                                                     GraphRecurrentNN)
    alegnn.modules.{model,training,evaluation}  ->  alegnn_amd.modules.{model,training,evaluation}  (Trainer with batch DP;
                                                    losses: any torch loss or the reference's own alegnn.modules.loss wrapper)
`install(reference_gml)` rebinds the reference's own symbols.
"""
from .functional import EVGF_edges, LSIGF
from .gso import EdgePattern, SparseGSO

__all__ = ["LSIGF", "EVGF_edges", "SparseGSO", "EdgePattern", "install"]
__version__ = "0.1.0"


def install(reference_graphML_module):
    """Monkey-patch the reference: ``import alegnn.utils.graphML as gml; alegnn_amd.install(gml)`` makes every
    architecture that instantiates ``gml.GraphFilter`` / calls ``gml.LSIGF`` (architectures.py:277, 665, 990, 4501, 4824;
    graphML.py:592, 1403, 1461) run on the HIP path."""
    from .utils import graphML as amd_gml
    reference_graphML_module.GraphFilter = amd_gml.GraphFilter
    reference_graphML_module.LSIGF = amd_gml.LSIGF
    reference_graphML_module.EdgeVariantGF = amd_gml.EdgeVariantGF      # architectures.py:1877, 2111
    reference_graphML_module.NVGF = amd_gml.NVGF
    reference_graphML_module.NodeVariantGF = amd_gml.NodeVariantGF      # architectures.py:1630
    reference_graphML_module.GatedGRNN = amd_gml.GatedGRNN              # graphML.py:3642, 3810, 3985, 4163 (the HiddenState family)
    reference_graphML_module.HiddenState = amd_gml.HiddenState          # architectures.py:4497
    reference_graphML_module.TimeGatedHiddenState = amd_gml.TimeGatedHiddenState      # architectures.py:4812
    reference_graphML_module.NodeGatedHiddenState = amd_gml.NodeGatedHiddenState      # architectures.py:4816
    reference_graphML_module.jARMA = amd_gml.jARMA                      # graphML.py:2826 (GraphFilterARMA.forward)
    reference_graphML_module.EdgeGatedHiddenState = amd_gml.EdgeGatedHiddenState      # architectures.py (GatedGCRNN edge gating)
    reference_graphML_module.LSIGF_DB = amd_gml.LSIGF_DB                # graphML.py:1164, 3366
    reference_graphML_module.GRNN_DB = amd_gml.GRNN_DB                  # graphML.py:3502
    reference_graphML_module.GraphFilter_DB = amd_gml.GraphFilter_DB    # architectures.py (LocalActivationGNN_DB / GraphRecurrentNN_DB)
    reference_graphML_module.HiddenState_DB = amd_gml.HiddenState_DB
    return reference_graphML_module
