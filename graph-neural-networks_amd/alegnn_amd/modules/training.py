"""Trainer / TrainerSingleNode with batch-axis data parallelism -- the reference's alegnn/modules/training.py:29-714.

Same constructor ``(model, data, nEpochs, batchSize, **kwargs)``, same options (doLogging, doSaveVars, printInterval,
learningRateDecayRate / learningRateDecayPeriod, validationInterval, earlyStoppingLag, graphNo, realizationNo), same
loop (random epoch permutation from numpy's global RNG, validation every ``validationInterval`` steps starting with step 0,
'Best' / 'Last' checkpoints, early stopping lag, 'Best' reloaded at the end) and the same ``trainVars`` dict, so
``Model(archit, loss, optim, Trainer, evaluate, ...)`` from the reference's examples (sourceLocGNN.py:760-772) trains unchanged.

Data parallelism (the reference has none): when ``torch.distributed`` is initialised every rank runs this same loop on an
identical replica.  Per step
    * the step's sample indices come from rank 0's permutation (broadcast once per epoch) and are cut into one contiguous
      share per rank (``trainBatch``, replaces training.py:398-399 handing the whole batch to one device);
    * each rank scales its share's mean loss by n_local * world / n_global, so the ONE all-reduce-mean of the flat gradient
      bucket (parallel.GradBucket) gives exactly the gradient of the global batch mean -- also for uneven shares;
    * the two reported scalars (loss, cost) ride in the bucket's tail, no second collective.
Validation is computed by every rank on the full validation set (identical replicas -> identical decisions, no exchange);
checkpoints are written by rank 0 (model.py).

``hipGraph=True`` (superset option, GPU only): zero-grad + forward + loss + backward of a training step are captured once per
batch size as a HIP graph and replayed; the samples are copied into the graph's static input buffers, the (all-reduce and)
optimizer step stay eager.  The reference's own configurations (N = 100 nodes, batch 20-100) are launch-bound -- about forty
small kernels per step -- and replay halves the step time; the arithmetic is the same kernels in the same
order, so the trajectory is bit-identical to the eager one.  The C library underneath only launches on the stream it is handed,
allocates nothing and never synchronises, which is what makes the step capturable.
"""
from __future__ import annotations

import os
import pickle
import time

import numpy as np
import torch
import torch.distributed as dist

from ..parallel import GradBucket, broadcast, broadcast_parameters


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class Trainer:
    def __init__(self, model, data, nEpochs, batchSize, **kwargs):
        self.model = model
        self.data = data
        self.rank, self.world = _world()

        doLogging = kwargs.get('doLogging', False)                  # training.py:91-165, same defaults
        doSaveVars = kwargs.get('doSaveVars', True)
        if 'printInterval' in kwargs:
            printInterval = kwargs['printInterval']
            doPrint = printInterval > 0
        else:
            doPrint = True
            printInterval = (data.nTrain // batchSize) // 5
        doLearningRateDecay = 'learningRateDecayRate' in kwargs and 'learningRateDecayPeriod' in kwargs
        validationInterval = kwargs.get('validationInterval', data.nTrain // batchSize)
        doEarlyStopping = 'earlyStoppingLag' in kwargs
        earlyStoppingLag = kwargs.get('earlyStoppingLag', 0)
        graphNo = kwargs.get('graphNo', -1)
        realizationNo = -1
        if 'realizationNo' in kwargs:
            if 'graphNo' in kwargs:
                realizationNo = kwargs['realizationNo']
            else:
                graphNo = kwargs['realizationNo']
        logger = None
        if doLogging:
            raise NotImplementedError("doLogging needs the reference's tensorboard Visualizer (visualTools.py); not rebuilt")
        if nEpochs == 0:
            doSaveVars = False
        if self.rank != 0:                                          # one rank talks and writes
            doPrint, doSaveVars = False, False

        nTrain = data.nTrain                                        # :173-195: batch sizes, the last one takes the remainder
        if nTrain < batchSize:
            sizes = [nTrain]
        else:
            sizes = [batchSize] * int(np.ceil(nTrain / batchSize))
            sizes[-1] -= sum(sizes) - nTrain
        nBatches = len(sizes)
        batchIndex = [0] + np.cumsum(sizes).tolist()

        self.trainingOptions = {
            'doLogging': doLogging, 'logger': logger, 'doSaveVars': doSaveVars, 'doPrint': doPrint,
            'printInterval': printInterval, 'doLearningRateDecay': doLearningRateDecay,
            'validationInterval': validationInterval, 'doEarlyStopping': doEarlyStopping,
            'earlyStoppingLag': earlyStoppingLag, 'batchIndex': batchIndex, 'batchSize': sizes, 'nEpochs': nEpochs,
            'nBatches': nBatches, 'graphNo': graphNo, 'realizationNo': realizationNo}
        if doLearningRateDecay:
            self.trainingOptions['learningRateDecayRate'] = kwargs['learningRateDecayRate']
            self.trainingOptions['learningRateDecayPeriod'] = kwargs['learningRateDecayPeriod']

        self.useGraph = bool(kwargs.get('hipGraph', False))
        self._graphs = {}                                           # batch size -> captured step
        self.bucket = None
        if self.world > 1:
            broadcast_parameters(self.model.archit)                 # identical replicas from step 0
            self.bucket = GradBucket(self.model.archit.parameters(), extra=2)

    # ---- one optimisation step ------------------------------------------------------------------------------------
    def _share(self, thisBatchIndices):
        """This rank's contiguous part of the step's indices (possibly empty when the batch is smaller than the world)."""
        if self.world == 1:
            return thisBatchIndices
        cuts = np.linspace(0, len(thisBatchIndices), self.world + 1).round().astype(np.int64)
        return thisBatchIndices[cuts[self.rank]:cuts[self.rank + 1]]

    def _forward(self, x, samplesType, indices):
        return self.model.archit(x)

    # ---- the step as a HIP graph (hipGraph=True) ---------------------------------------------------------------------
    def _graph_step(self, xTrain, yTrain, samplesType, indices, weight):
        """zero-grad + forward + loss + backward for this batch shape, captured on first use and replayed afterwards.
        Returns (loss tensor, output tensor): the graph's static outputs, valid until the next replay."""
        # A captured step embeds raw device pointers of the GSO plans and the per-rank loss weight: both are part of the key, and the
        # entry keeps the GSO objects alive (their finalizer frees the plans) -- changeGSO() simply leads to a new capture.
        gsos = tuple(m._gso for m in self.model.archit.modules() if getattr(m, "_gso", None) is not None)
        gso_ids = tuple(id(g_) for g_ in gsos)
        key = (tuple(xTrain.shape), tuple(yTrain.shape), xTrain.dtype, yTrain.dtype, float(weight), gso_ids)
        g = self._graphs.get(key)
        if g is None:
            # captures taken for GSOs the architecture no longer uses (changeGSO) would pin their static buffers and device plans for
            # the life of the trainer: drop them; the few batch shapes of the current GSO set stay (bounded: oldest first beyond 8)
            for k in [k for k in self._graphs if k[5] != gso_ids]:
                del self._graphs[k]
            while len(self._graphs) >= 8:
                del self._graphs[next(iter(self._graphs))]
            dev = self.model.device
            sx = torch.empty(xTrain.shape, dtype=xTrain.dtype, device=dev)
            sy = torch.empty(yTrain.shape, dtype=yTrain.dtype, device=dev)
            sx.copy_(xTrain)
            sy.copy_(yTrain)
            params = [p for p in self.model.archit.parameters() if p.requires_grad]

            def body():
                if self.bucket is None:
                    for p in params:
                        p.grad = None                               # the captured backward then writes (not accumulates) the grads
                else:
                    self.bucket.zero_()
                yHat = self._forward(sx, samplesType, indices)
                lossValue = self.model.loss(yHat, sy)
                (lossValue * weight if self.world > 1 else lossValue).backward()
                return lossValue, yHat

            side = torch.cuda.Stream(device=dev)                    # warm-up off the capture: plans, LDS attributes, allocator pools
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    body()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = body()
            g = self._graphs[key] = dict(graph=graph, x=sx, y=sy, out=out, weight=weight, gsos=gsos,
                                         grads=[p.grad for p in params], params=params)
        g["x"].copy_(xTrain, non_blocking=True)
        g["y"].copy_(yTrain, non_blocking=True)
        g["graph"].replay()
        if self.bucket is None:
            for p, gr in zip(g["params"], g["grads"]):
                p.grad = gr                                         # an eager step in between may have re-pointed .grad
        return g["out"]

    def trainBatch(self, thisBatchIndices):
        mine = self._share(thisBatchIndices)
        startTime = time.perf_counter()
        if self.useGraph and len(mine) > 0 and torch.device(self.model.device).type == "cuda":
            xTrain, yTrain = self.data.getSamples('train', mine)
            weight = len(mine) * self.world / len(thisBatchIndices)
            # single-node trainers look the target nodes up per batch: their ids are baked into a captured step, so those stay eager
            if type(self)._forward is Trainer._forward:
                lossValueTrain, yHatTrain = self._graph_step(xTrain, yTrain, 'train', mine, weight)
                lossValue = lossValueTrain.item() * weight
                costValue = float(self.data.evaluate(yHatTrain.data, yTrain.to(self.model.device))) * weight
                if self.bucket is not None:
                    tail = self.bucket.extra
                    tail[0], tail[1] = lossValue, costValue
                    self.bucket.allreduce_mean()
                    lossValue, costValue = float(tail[0]), float(tail[1])
                self.model.optim.step()
                return lossValue, costValue, time.perf_counter() - startTime
        if self.bucket is None:
            self.model.archit.zero_grad()                           # :241
        else:
            self.bucket.zero_()
        lossValue = costValue = 0.0
        if len(mine) > 0:
            xTrain, yTrain = self.data.getSamples('train', mine)
            xTrain = xTrain.to(self.model.device)
            yTrain = yTrain.to(self.model.device)
            yHatTrain = self._forward(xTrain, 'train', mine)        # :244
            lossValueTrain = self.model.loss(yHatTrain, yTrain)     # :247
            weight = len(mine) * self.world / len(thisBatchIndices)
            (lossValueTrain * weight if self.world > 1 else lossValueTrain).backward()      # :250
            lossValue = lossValueTrain.item() * weight
            costValue = float(self.data.evaluate(yHatTrain.data, yTrain)) * weight
        if self.bucket is not None:
            tail = self.bucket.extra
            tail[0], tail[1] = lossValue, costValue
            self.bucket.allreduce_mean()                            # gradients + the two scalars, one collective
            lossValue, costValue = float(tail[0]), float(tail[1])
        self.model.optim.step()                                     # :253
        return lossValue, costValue, time.perf_counter() - startTime

    def validationStep(self):
        xValid, yValid = self.data.getSamples('valid')
        xValid = xValid.to(self.model.device)
        yValid = yValid.to(self.model.device)
        startTime = time.perf_counter()
        with torch.no_grad():
            yHatValid = self._forward(xValid, 'valid', None)        # :282
            lossValueValid = self.model.loss(yHatValid, yValid)
            timeElapsed = time.perf_counter() - startTime
            costValid = self.data.evaluate(yHatValid, yValid)
        return lossValueValid.item(), float(costValid), timeElapsed

    # ---- the loop -------------------------------------------------------------------------------------------------
    def _epoch_order(self):
        """np.random.permutation(nTrain) as at training.py:379; under DP rank 0's draw is used by every rank."""
        order = np.random.permutation(self.data.nTrain)
        if self.world > 1:
            t = torch.as_tensor(order, dtype=torch.int64, device=self.bucket.flat.device)
            broadcast(t, src=0)
            order = t.cpu().numpy()
        return [int(i) for i in order]

    def train(self):
        opt = self.trainingOptions
        doPrint, printInterval = opt['doPrint'], opt['printInterval']
        doEarlyStopping, earlyStoppingLag = opt['doEarlyStopping'], opt['earlyStoppingLag']
        batchIndex, nBatches, nEpochs = opt['batchIndex'], opt['nBatches'], opt['nEpochs']
        validationInterval = opt['validationInterval']
        tag = ""
        if opt['graphNo'] > -1:
            tag = "%d" % opt['graphNo'] + ("/%d" % opt['realizationNo'] if opt['realizationNo'] > -1 else "")
        scheduler = None
        if opt['doLearningRateDecay']:                              # :350-352
            scheduler = torch.optim.lr_scheduler.StepLR(self.model.optim, opt['learningRateDecayPeriod'],
                                                        opt['learningRateDecayRate'])
        lossTrain, costTrain, lossValid, costValid, timeTrain, timeValid = [], [], [], [], [], []
        epoch = lagCount = 0
        bestScore, bestEpoch, bestBatch, initialBest = None, 0, 0, True

        def keepGoing():
            return lagCount < earlyStoppingLag or not doEarlyStopping

        while epoch < nEpochs and keepGoing():
            idxEpoch = self._epoch_order()
            if scheduler is not None:
                scheduler.step()                                    # stepped at the START of the epoch, as at :385
                if doPrint:
                    print("Epoch %d, learning rate = %.8f" % (epoch + 1, self.model.optim.param_groups[0]['lr']))
            batch = 0
            while batch < nBatches and keepGoing():
                step = epoch * nBatches + batch
                lossValue, costValue, timeElapsed = self.trainBatch(idxEpoch[batchIndex[batch]:batchIndex[batch + 1]])
                lossTrain.append(lossValue)
                costTrain.append(costValue)
                timeTrain.append(timeElapsed)
                if doPrint and printInterval > 0 and step % printInterval == 0:
                    print("\t(E: %2d, B: %3d) %6.4f / %7.4f - %6.4fs" % (epoch + 1, batch + 1, costValue, lossValue,
                                                                        timeElapsed), "[%s]" % tag if tag else "")
                if step % validationInterval == 0:                  # :436
                    lossValue, costValue, timeElapsed = self.validationStep()
                    lossValid.append(lossValue)
                    costValid.append(costValue)
                    timeValid.append(timeElapsed)
                    if doPrint:
                        print("\t(E: %2d, B: %3d) %6.4f / %7.4f - %6.4fs [VALIDATION%s (%s)]" % (
                            epoch + 1, batch + 1, costValue, lossValue, timeElapsed, "." + tag if tag else "",
                            self.model.name))
                    if bestScore is None:                           # first validation: :478-485
                        bestScore, bestEpoch, bestBatch = costValue, epoch, batch
                        self.model.save(label='Best')
                    elif costValue < bestScore:                     # :488-500
                        bestScore, bestEpoch, bestBatch = costValue, epoch, batch
                        if doPrint:
                            print("\t=> New best achieved: %.4f" % bestScore)
                        self.model.save(label='Best')
                        initialBest = False
                        lagCount = 0
                    elif doEarlyStopping and not initialBest:       # :503-504
                        lagCount += 1
                batch += 1
            epoch += 1

        self.model.save(label='Last')                               # :516
        trainVars = {'nEpochs': nEpochs, 'nBatches': nBatches, 'validationInterval': validationInterval,
                     'batchSize': np.array(opt['batchSize']), 'batchIndex': np.array(batchIndex),
                     'lossTrain': np.array(lossTrain), 'costTrain': np.array(costTrain),
                     'lossValid': np.array(lossValid), 'costValid': np.array(costValid)}
        if opt['doSaveVars']:                                       # :541-548
            saveDirVars = os.path.join(self.model.saveDir, 'trainVars')
            os.makedirs(saveDirVars, exist_ok=True)
            with open(os.path.join(saveDirVars, self.model.name + 'trainVars.pkl'), 'wb') as f:
                pickle.dump(trainVars, f)
        if nEpochs == 0:                                            # :554-559
            self.model.save(label='Best')
            self.model.save(label='Last')
            if doPrint:
                print("WARNING: No training. Best and Last models are the same.")
        self.model.load(label='Best')                               # :563
        if doPrint and nEpochs > 0:
            print("=> Best validation achieved (E: %d, B: %d): %.4f" % (bestEpoch + 1, bestBatch + 1, bestScore))
        return trainVars


class TrainerSingleNode(Trainer):
    """Loss at one target node per sample (MovieLens, training.py:580-714): the architecture needs ``singleNodeForward``
    (LocalGNN), the data a ``getLabelID(samplesType[, indices])``."""

    def __init__(self, model, data, nEpochs, batchSize, **kwargs):
        assert 'singleNodeForward' in dir(model.archit)             # :640-641
        assert 'getLabelID' in dir(data)
        super().__init__(model, data, nEpochs, batchSize, **kwargs)

    def _forward(self, x, samplesType, indices):
        targetIDs = self.data.getLabelID(samplesType) if indices is None else self.data.getLabelID(samplesType, indices)
        return self.model.archit.singleNodeForward(x, targetIDs)    # :660, :698
