"""TrainingWorker with the reference's constructor and process_batch() contract
(alphazero/training.py:15-125): SGD + momentum + weight decay, MultiStepLR, masked
cross-entropy + MSE, weights published after every step.  On the GPU the loss with its gradients and the
optimiser update are hand-written kernels (loss.AlphaZeroLoss, optim.FlatSGD; csrc/az_train.hip).  For the
Connect Four network with 32 trunk channels the whole pass -- forward on the batch's statistics, loss, backward,
running statistics -- is hand-written too (train_c4.C4TrainStep, csrc/az_train_c4.hip: one resident launch,
no vendor convolution, no autograd; AZ_TRAIN_HANDWRITTEN=0 switches it off); other networks go through PyTorch's
forward and backward.  With a DeviceReplayWindow on the GPU the whole step is recorded once as a HIP graph after
three ordinary steps and replayed from then on, the batch gathered into fixed tensors and the learning rate read
from device memory (AZ_TRAIN_GRAPH=0 or record_step=False keeps every step eager).  The summary
writer and model server are duck-typed (`add_scalar`, `update(model, version)`).  `dataset` is a
ReplayBufferDataset (the reference's per-state sample dicts) or a DeviceReplayWindow (batches taken
straight from tensors on the GPU)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch.optim import SGD
from torch.optim.lr_scheduler import MultiStepLR

from .loss import masked_cross_entropy, masked_kl_div
from .replay_buffer import ReplayBufferSampler


class TrainingWorker:
    EAGER_STEPS = 3       # ordinary steps before the step is recorded (the libraries pick their kernels in them)
    CHECK_EVERY = 16      # recorded steps between two read-backs of the target-distribution check

    def __init__(self, config, model_server, dataset, summary_writer, device, num_workers=0, grad_sync=None,
                 batch_size=None, record_step=None):
        self._config, self._model_server, self._dataset = config, model_server, dataset
        # data-parallel use: grad_sync(model) runs between backward() and the optimizer step (e.g.
        # distributed.allreduce_gradients) and batch_size is this rank's share of config's batch
        self._grad_sync = grad_sync
        self._batch_size = int(batch_size if batch_size is not None else config.training['batch_size'])
        self._writer, self._device, self._num_workers = summary_writer, device, num_workers
        self._model = config.Model().to(device)
        self._loss = config.Loss()
        t = config.training
        if torch.device(device).type == 'cuda':
            # one launch per step over a flat parameter buffer (csrc/az_train.hip)
            from .optim import FlatSGD
            self._optimizer = FlatSGD(self._model.parameters(), lr=t['lr'], momentum=t['momentum'],
                                      weight_decay=t['weight_decay'])
        else:
            self._optimizer = SGD(self._model.parameters(), lr=t['lr'], momentum=t['momentum'],
                                  weight_decay=t['weight_decay'])
        self._lr_schedule = MultiStepLR(self._optimizer, t['lr_schedule'], gamma=t['lr_schedule_gamma'])
        self._sampler = ReplayBufferSampler(dataset, batch_size=self._batch_size, bufferlen=64 * 1024)
        self._iter = 0
        import os
        if record_step is None:
            record_step = os.environ.get('AZ_TRAIN_GRAPH', '1') != '0'
        self._record = bool(record_step and torch.device(device).type == 'cuda' and hasattr(dataset, 'sample_into')
                            and grad_sync is None and hasattr(self._optimizer, 'step_recorded'))
        self._graph = None
        # batches from a DeviceReplayWindow are drawn and gathered by one kernel (az_sample_window): a function of a
        # seed taken from torch's generator here and of a draw counter in device memory, so a recorded step can
        # contain its own sampling
        self._draw = None
        if (torch.device(device).type == 'cuda' and hasattr(dataset, 'kernel_sampling')
                and dataset.kernel_sampling(self._batch_size)):
            self._sample_seed = None           # drawn from torch's generator at the first batch (see _seed())
            self._draw = torch.zeros(1, dtype=torch.int64, device=device)
        # the hand-written pass binds the parameters and their .grad tensors where FlatSGD has put them
        self._hand = None
        if torch.device(device).type == 'cuda' and os.environ.get('AZ_TRAIN_HANDWRITTEN', '1') != '0' and self._batch_size >= 2:
            from . import train_c4
            from .loss import AlphaZeroLoss
            if train_c4.supports(self._model) and type(self._loss) is AlphaZeroLoss:
                self._hand = train_c4.C4TrainStep(self._model)

    @property
    def model(self):
        return self._model

    def _seed(self):
        # taken when the first batch is drawn, not at construction: a data-parallel job seeds every rank alike for
        # the initial weights and differently afterwards for the sample streams (train.synchronous_play_and_train)
        if self._sample_seed is None:
            self._sample_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return self._sample_seed

    def _batch(self):
        if hasattr(self._dataset, 'sample'):
            # DeviceReplayWindow: the window already lives on the GPU as tensors
            if self._draw is not None:
                b = self._dataset.sample(self._batch_size, seed=self._seed(), draw_dev=self._draw)
            else:
                b = self._dataset.sample(self._batch_size)
            return (b['x'].to(self._device), b['p'].to(self._device), b['z'].to(self._device),
                    b['p_valid'].to(self._device), None)
        items = [self._dataset[i] for i in next(self._sampler)]
        stack = lambda k, dt: torch.as_tensor(np.stack([it[k] for it in items])).to(self._device, dt)
        solved = torch.as_tensor([bool(it['solved']) for it in items])
        extra = None
        if bool(solved.any()):
            extra = {'solved': solved.to(self._device), 'v*': stack('v*', torch.float32),
                     'move_scores': stack('move_scores', torch.float32)}
        return (stack('x', torch.float32), stack('p', torch.float32), stack('z', torch.float32),
                stack('p_valid', torch.bool), extra)

    def _log_solved(self, extra, p, z, p_hat, v_hat, p_valid, it):
        """Distance of the search targets and of the predictions from the solver's answers on the
        solved states of the batch (alphazero/training.py:85-113)."""
        w = self._writer
        with torch.no_grad():
            solved = extra['solved']
            v_star = extra['v*'][solved]
            scores = extra['move_scores'][solved].clone()
            scores[~p_valid[solved]] = -float('inf')
            p_star = F.softmax(scores, dim=1)
            logp = torch.log(p)
            w.add_scalar('eval/policy_target_KL', masked_kl_div(logp[solved], p_star, p_valid[solved]).mean(), it)
            w.add_scalar('eval/policy_prediction_KL', masked_kl_div(p_hat[solved], p_star, p_valid[solved]).mean(), it)
            w.add_scalar('eval/value_target_mse', ((v_star - z[solved]) ** 2).mean(), it)
            w.add_scalar('eval/value_prediction_mse', ((v_star - v_hat[solved]) ** 2).mean(), it)
            w.add_scalar('eval/policy_target_entropy', masked_cross_entropy(logp, p, p_valid).mean(), it)
            w.add_scalar('eval/policy_prediction_entropy',
                         masked_cross_entropy(p_hat, F.softmax(p_hat, dim=1), p_valid).mean(), it)

    def process_batch(self):
        N, B = len(self._dataset), self._batch_size
        M = max(2 * self._num_workers, 1) * B
        if N < M:
            print(f'Skipping batch, too few examples in replay buffer ({N} < {M}).')
            return 0
        if not self._model.training:                   # (walking the module tree costs 0.1 ms: only when it changes anything)
            self._model.train()
        if self._record and self._iter >= self.EAGER_STEPS:
            return self._process_batch_recorded()
        x, p, z, p_valid, extra = self._batch()
        if self._hand is not None:
            # forward, loss and backward in the hand-written kernels: the gradients land in the .grad tensors
            out3 = self._hand.run(x.contiguous(), p.contiguous(), z.contiguous(), p_valid.contiguous())
            policy_loss, value_loss = out3[0], out3[1]
            assert float(out3[2]) < 1e-6, f'Target policy must be a valid distribution: {p}'       # loss.py:58-59
            total = policy_loss + value_loss
            p_hat, v_hat = self._hand.logits, self._hand.v_hat.reshape(-1, 1)
        else:
            p_hat, v_hat = self._model(x, p_valid)
            policy_loss, value_loss = self._loss(p, z, p_hat, v_hat, p_valid)
            self._optimizer.zero_grad()
            total = policy_loss + value_loss
            total.backward()
        if self._grad_sync is not None:
            self._grad_sync(self._model)
        self._optimizer.step()
        self._lr_schedule.step()
        it = self._iter
        self._iter += 1
        self._model_server.update(self._model, it + 1)
        if self._writer is not None:
            self._writer.add_scalar('loss/policy', policy_loss, it)
            self._writer.add_scalar('loss/value', value_loss, it)
            self._writer.add_scalar('loss/total', total, it)
            self._writer.add_scalar('optim/learning_rate', self._lr_schedule.get_last_lr()[0], it)
            if extra is not None:
                self._log_solved(extra, p, z, p_hat, v_hat, p_valid, it)
            db = getattr(self._dataset, 'db', None)
            if db is not None:                                     # training.py:117-123
                self._writer.add_scalar('data/total_games', db.num_games(), it)
                self._writer.add_scalar('data/distinct_games', db.num_distinct_games(), it)
                self._writer.add_scalar('data/total_states', db.num_states(), it)
        return it + 1

    # ---- the step as one recorded launch sequence ------------------------------------------------------------
    def _record_step(self):
        ds, B, dev = self._dataset, self._batch_size, self._device
        self._gx = torch.empty((B,) + tuple(ds.x.shape[1:]), dtype=ds.x.dtype, device=dev)
        self._gp = torch.empty((B,) + tuple(ds.p.shape[1:]), dtype=ds.p.dtype, device=dev)
        self._gz = torch.empty((B,) + tuple(ds.z.shape[1:]), dtype=ds.z.dtype, device=dev)
        self._gvalid = torch.empty((B,) + tuple(ds.p_valid.shape[1:]), dtype=ds.p_valid.dtype, device=dev)
        in_graph = self._draw is not None
        seed = self._seed() if in_graph else None
        if not in_graph:
            ds.sample_into(self._gx, self._gp, self._gz, self._gvalid)
        self._sample_in_graph = in_graph
        self._optimizer.set_recorded_lr()
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if in_graph:
                ds.sample_into(self._gx, self._gp, self._gz, self._gvalid, seed=seed, draw_dev=self._draw)
            if self._hand is not None:
                out3 = self._hand.run(self._gx, self._gp, self._gz, self._gvalid)
                policy_loss, value_loss, guard = out3[0], out3[1], out3[2:3]
                total = None                           # (added up when somebody logs it: one launch less per step)
            else:
                p_hat, v_hat = self._model(self._gx, self._gvalid)
                policy_loss, value_loss = self._loss(self._gp, self._gz, p_hat, v_hat, self._gvalid)
                self._optimizer.zero_grad()
                total = policy_loss + value_loss
                total.backward()
                guard = getattr(self._loss, 'target_deviation', None)
            # the update itself is conditional on the loss kernel's check of the targets, on the device: a batch
            # that fails it changes no weight (the reference asserts before optimizer.step(), loss.py:58-59)
            self._optimizer.step_recorded(guard=guard)
        self._graph = graph
        self._g_out = (policy_loss.detach(), value_loss.detach(), None if total is None else total.detach())
        return graph

    def _process_batch_recorded(self):
        if self._graph is None:
            self._record_step()                       # (without in-graph sampling its batch is already in place)
        elif not self._sample_in_graph:
            self._dataset.sample_into(self._gx, self._gp, self._gz, self._gvalid)
        self._optimizer.set_recorded_lr()
        self._graph.replay()
        self._optimizer._opt_called = True            # what MultiStepLR looks at before it steps
        self._lr_schedule.step()
        it = self._iter
        self._iter += 1
        self._model_server.update(self._model, it + 1)
        if self._writer is not None or it % self.CHECK_EVERY == 0:
            # Read back every CHECK_EVERY-th step (every step with a writer).  A batch that failed the check in
            # between had its update withheld by the recorded step itself (the weights published for it are the
            # previous ones), so nothing wrong was ever applied or handed out; the assertion just comes late.
            withheld = int(self._optimizer.withheld)
            assert withheld == 0, f'Target policy must be a valid distribution ({withheld} update(s) withheld)'
        if self._writer is not None:
            policy_loss, value_loss, total = self._g_out
            if total is None:
                total = policy_loss + value_loss
            self._writer.add_scalar('loss/policy', policy_loss, it)
            self._writer.add_scalar('loss/value', value_loss, it)
            self._writer.add_scalar('loss/total', total, it)
            self._writer.add_scalar('optim/learning_rate', self._lr_schedule.get_last_lr()[0], it)
        return it + 1
