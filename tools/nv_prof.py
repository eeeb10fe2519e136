import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/graph-neural-networks_amd"]
import torch
from alegnn_amd import graphgen
from alegnn_amd.utils import graphML as gml
dev = torch.device("cuda:0")
N, B, G, F, K, M = 10000, 64, 32, 32, 5, 1000
A = graphgen.sbm(N, avg_degree=10.0, seed=0)
layer = gml.NodeVariantGF(G, F, K, M, 1, True); layer.addGSO(A); layer.to(dev)
x = torch.randn(B, G, N, device=dev, requires_grad=True)
for _ in range(5):
    layer.zero_grad(set_to_none=True); x.grad = None
    layer(x).square().sum().backward()
torch.cuda.synchronize()
