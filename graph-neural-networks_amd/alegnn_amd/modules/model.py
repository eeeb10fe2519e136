"""Model: architecture + loss + optimizer + trainer + evaluator under one name -- the reference's alegnn/modules/model.py:14-136.

Same constructor and methods (``train``, ``evaluate``, ``save``, ``load``, ``getTrainingOptions``), same checkpoint files
(``<saveDir>/savedModels/<name>Archit<label>.ckpt`` and ``...Optim<label>.ckpt``, each a ``torch.save`` of a state_dict,
model.py:106-129): checkpoints written by the reference load here and the other way round (tests/test_host_logic.py).

Under ``torch.distributed`` (batch-axis data parallelism, parallel.py) every rank holds an identical replica, so only rank 0
writes; ``load`` waits for that write.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class Model:
    def __init__(self, architecture, loss, optimizer, trainer, evaluator, device, name, saveDir):
        self.archit = architecture
        self.archit.to(device)                                      # model.py:69
        self.nParameters = sum(p.numel() for p in self.archit.parameters() if p.dim() > 0)      # :71-79
        self.loss = loss
        self.optim = optimizer
        self.trainer = trainer
        self.evaluator = evaluator
        self.device = device
        self.name = name
        self.saveDir = saveDir

    def train(self, data, nEpochs, batchSize, **kwargs):
        self.trainer = self.trainer(self, data, nEpochs, batchSize, **kwargs)      # :97
        return self.trainer.train()

    def evaluate(self, data, **kwargs):
        return self.evaluator(self, data, **kwargs)

    def _files(self, label, saveDir):
        stem = os.path.join(saveDir, 'savedModels', self.name)
        return stem + 'Archit' + label + '.ckpt', stem + 'Optim' + label + '.ckpt'

    def save(self, label='', **kwargs):
        saveDir = kwargs['saveDir'] if 'saveDir' in kwargs.keys() else self.saveDir
        architFile, optimFile = self._files(label, saveDir)
        rank, world = _rank_world()
        if rank == 0:
            os.makedirs(os.path.dirname(architFile), exist_ok=True)
            torch.save(self.archit.state_dict(), architFile)        # :116-117
            torch.save(self.optim.state_dict(), optimFile)
        if world > 1:
            dist.barrier()

    def load(self, label='', **kwargs):
        if 'loadFiles' in kwargs.keys():
            architFile, optimFile = kwargs['loadFiles']             # :120-121
        else:
            architFile, optimFile = self._files(label, self.saveDir)
        # in place: parameters keep their storage (GradBucket views and captured HIP graphs stay valid)
        self.archit.load_state_dict(torch.load(architFile, map_location=self.device))
        self.optim.load_state_dict(torch.load(optimFile, map_location=self.device))

    def getTrainingOptions(self):
        return self.trainer.trainingOptions if 'trainingOptions' in dir(self.trainer) else None

    def __repr__(self):
        return "Name: %s\nNumber of learnable parameters: %d\n\nModel architecture:\n----- -------------\n%r\n\n" \
               "Loss function:\n---- ---------\n%r\n\nOptimizer:\n----------\n%r\n" \
               "Training algorithm:\n-------- ----------\n%r\nEvaluation algorithm:\n---------- ----------\n%r\n" % (
                   self.name, self.nParameters, self.archit, self.loss, self.optim, self.trainer, self.evaluator)
