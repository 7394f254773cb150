"""Command-line front end with the reference's flag surface (/root/reference/main.py:31-52), verbatim.

    python -m graphpope_amd.main --dataset flickr --embedding_space geodesic --sampling_method stochastic \
        --num_anchor_nodes 256 --num_layers 3 --epochs 5

Same 15 flags, names, types and defaults -- including ``--wandb_logging`` being ``type=bool`` (any
non-empty string is True, main.py:49) and ``--dropout`` being parsed but never handed to the model
(main.py:272).  ``--embedding_space baseline`` skips GraphPOPE (main.py:94).

What differs, because PyG / Lightning / the datasets are not available offline: the graph comes from
``<--data_dir or ./data>/<dataset>.npz`` (arrays ``x, y, edge_index, train_mask, val_mask, test_mask``) if present
and otherwise from a synthetic graph of the dataset's shape; the Lightning ``Trainer`` is a plain loop with the
same optimiser, scheduler, clipping and early stopping (main.py:243-255, 279-290).  Flags are parsed in ``main()``
rather than at import.
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import synth
from .optim import Adam
from . import engine
from .sage import SAGE, IndexedFeatures, cross_entropy
from .sampler import NeighborSampler
from .train import SageTrainStep
from .utils import Graphpope


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description='GraphPOPE')
    # Pope arguments
    parser.add_argument('--dataset', type=str, default='flickr')  # flickr, pubmed
    parser.add_argument('--embedding_space', type=str, default='geodesic')  # node2vec, geodesic, baseline
    parser.add_argument('--sampling_method', type=str, default='degree_centrality')
    parser.add_argument('--num_anchor_nodes', type=int, default=2)
    parser.add_argument('--distance_function', type=str, default=None)  # distance, similarity, euclidean
    parser.add_argument('--num_workers', type=int, default=6)
    # Additional hyperparams
    parser.add_argument('--dropout', type=float, default=0.5)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--num_layers', type=int, default=3)
    parser.add_argument('--hidden_layer_size', type=int, default=256)
    parser.add_argument('--batch_size', type=int, default=1550)
    parser.add_argument('--epochs', type=int, default=300)
    parser.add_argument('--seed', type=int, default=42)
    parser.add_argument('--wandb_logging', type=bool, default=False)
    parser.add_argument('--n_gpus', type=int, default=1)
    return parser


class GraphData:
    """The attributes of a PyG ``Data`` object this path touches."""

    def __init__(self, x, y, edge_index, train_mask, val_mask, test_mask):
        self.x, self.y, self.edge_index = x, y, edge_index
        self.train_mask, self.val_mask, self.test_mask = train_mask, val_mask, test_mask
        self.num_nodes = int(x.shape[0])


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)                     # the anchor draw uses this global legacy stream (utils.py:22-24)
    torch.manual_seed(seed)


def load_dataset(name: str, data_dir: str) -> tuple[GraphData, int]:
    """(data, num_classes): real arrays if ``<data_dir>/<name>.npz`` exists, else a synthetic graph of that shape."""
    classes = 7 if name == 'flickr' else 3                                   # main.py:81-83, 141-143
    path = osp.join(data_dir, f'{name}.npz')
    if osp.exists(path):
        z = np.load(path)
        t = {k: torch.as_tensor(z[k]) for k in z.files}
        return GraphData(t['x'].float(), t['y'].long(), t['edge_index'].long(), t['train_mask'].bool(),
                         t['val_mask'].bool(), t['test_mask'].bool()), classes
    ei, n = synth.flickr_like() if name == 'flickr' else synth.pubmed_like()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, 500, generator=g)                                     # 500 features: main.py:77-79, 137-139
    y = torch.randint(0, classes, (n,), generator=g)
    split = torch.rand(n, generator=g)
    return GraphData(x, y, torch.as_tensor(ei), split < 0.5, (split >= 0.5) & (split < 0.75), split >= 0.75), classes


def _epoch_order(node_idx, shuffle, gen):
    return node_idx[torch.randperm(node_idx.numel(), device=node_idx.device, generator=gen)] if shuffle else node_idx


def _run_epoch(model, feats, labels, sampler, node_idx, args, gen, epoch, opt=None, trainer=None):
    """One pass over node_idx.  Everything stays on the device: the fan-out sampler (main.py:100-116), the feature
    gather of convert_batch (main.py:118-123), the model, the optimiser; loss / accuracy are accumulated on the
    device and read once per epoch.  Full training batches go through `trainer` (graphpope_amd.train.SageTrainStep:
    sampled with device extents, the whole step replayed as a HIP graph, nothing read back -- in its epoch mode the step
    also takes its seeds from the epoch's shuffled order and gathers their labels itself, so a training step has no
    per-batch input at all); the last, shorter batch of an epoch and the evaluation passes take the eager path below."""
    train = opt is not None
    model.train(train)
    dev = feats.device
    tot_loss = torch.zeros((), device=dev)
    tot_correct = torch.zeros((), device=dev, dtype=torch.int64)
    tot = 0
    order = _epoch_order(node_idx, train, gen)                                           # NeighborSampler(node_idx, shuffle=train)
    use_trainer = train and trainer is not None
    if use_trainer:
        trainer.set_epoch(order, labels)
    for b, lo in enumerate(range(0, order.numel(), args.batch_size)):
        seeds = order[lo:lo + args.batch_size]
        if use_trainer and seeds.numel() == args.batch_size:
            loss = trainer.step_epoch()                                                  # seeds = order[lo : lo + batch_size], y gathered on the way
            y_hat, y = trainer.logits, trainer.y
        else:
            y = labels.index_select(0, seeds)                                            # Batch.y = data.y[n_id[:batch_size]]
            n_id, adjs = sampler.sample(seeds, seed=(args.seed << 20) + (epoch << 10) + b)   # NeighborSampler(sizes=[25, 10])
            x = IndexedFeatures(feats, n_id)                                             # Batch.x = data.x[n_id], never materialised
            with torch.set_grad_enabled(train):
                y_hat = model(x, adjs)
                loss = cross_entropy(y_hat, y)                                           # main.py:216 F.cross_entropy
            if train:
                for p in model.parameters():
                    p.grad = None
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)                  # gradient_clip_val=0.5 (main.py:286)
                opt.step()
                if trainer is not None:
                    trainer.state.advance()                                              # the optimiser reads its step count from the trainer's device word
        tot_loss += loss.detach() * seeds.numel()
        tot_correct += (y_hat.argmax(-1) == y).sum()
        tot += seeds.numel()
    return float(tot_loss) / max(tot, 1), int(tot_correct) / max(tot, 1)


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    seed_everything(args.seed)
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    data_dir = os.environ.get('GRAPHPOPE_DATA_DIR', osp.join(os.getcwd(), 'data'))
    data, num_classes = load_dataset(args.dataset, data_dir)
    if args.embedding_space != 'baseline':                                               # main.py:94-98
        data.x = Graphpope(data=data, dataset=args.dataset, embedding_space=args.embedding_space,
                           sampling_method=args.sampling_method, num_anchor_nodes=args.num_anchor_nodes,
                           distance_function=args.distance_function, num_workers=args.num_workers)
    in_channels = int(500 + args.num_anchor_nodes)                                       # main.py:77-79 (hard-coded 500 + K)
    model = SAGE(in_channels, num_classes, args.hidden_layer_size, args.num_layers).to(dev)   # dropout NOT passed: main.py:272
    opt = Adam(model.parameters(), lr=args.lr)                                           # main.py:244 torch.optim.Adam rule, one launch per step
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt)                              # monitors val_loss
    # HBM-resident training data: features (+ POPE columns), labels, and adj_t as a device CSR whose row v lists the nodes v
    # aggregates from (main.py:84 ToSparseTensor: the transposed adjacency; the same CSR for the symmetric Flickr / PubMed)
    feats, labels = data.x.to(dev, torch.float32).contiguous(), data.y.to(dev)
    adj_t = engine.build_csr(data.edge_index.flip(0).contiguous().to(dev), data.num_nodes)
    sampler = NeighborSampler(adj_t.rowptr, adj_t.col, data.num_nodes, sizes=(25, 10))
    gen = torch.Generator(device=dev).manual_seed(args.seed)
    idx = {k: torch.nonzero(getattr(data, f'{k}_mask'), as_tuple=False).flatten().to(dev) for k in ('train', 'val', 'test')}
    best, bad = -1.0, 0
    torch.autograd.set_multithreading_enabled(False)          # backward in the calling thread: the step is launch-bound on the host
    trainer = None
    if os.environ.get('GRAPHPOPE_TRAIN_STEP', 'graph') != 'eager' and idx['train'].numel() >= args.batch_size:
        trainer = SageTrainStep(model, opt, feats, args.batch_size, sampler=sampler, clip=0.5, seed=args.seed)   # gradient_clip_val=0.5 (main.py:286)
    for epoch in range(args.epochs):
        tr_loss, tr_acc = _run_epoch(model, feats, labels, sampler, idx['train'], args, gen, epoch, opt, trainer)
        va_loss, va_acc = _run_epoch(model, feats, labels, sampler, idx['val'], args, gen, epoch)
        sched.step(va_loss)
        print(f'epoch {epoch}: train_loss {tr_loss:.4f} train_acc {tr_acc:.4f} val_loss {va_loss:.4f} val_acc {va_acc:.4f}')
        if va_acc > best:
            best, bad = va_acc, 0
        else:
            bad += 1
            if bad >= 20:                                                                # EarlyStopping(val_acc, patience=20)
                break
    _, te_acc = _run_epoch(model, feats, labels, sampler, idx['test'], args, gen, args.epochs)
    print(f'test_acc {te_acc:.4f}')
    return te_acc


if __name__ == "__main__":
    main()
