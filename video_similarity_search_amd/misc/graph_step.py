"""A whole training step — forward, loss, backward, optimizer update — captured once into a hipGraph and replayed.

The reference's loop issues its step from Python every iteration (online_train.py:136-230, 255-392); so does this package, and at small
batches that is what a step costs: at 8 clips of 3 x 16 x 112 x 112 (north_star's batch) the step is ~330 launches of 10-40 us kernels, issued
through autograd, ctypes and two streams — the GPU waits for the host.  The engine owns every buffer of a pass and its shapes are static
for a fixed batch, so the launch sequence of one step IS the step: `GraphedStep` runs the callable a few times eagerly (lazy set-up:
plans, workspaces, kernel attributes), captures one more run on a capture stream (the side stream's weight gradients and weight packs
fork from and join it through events, so they are part of the graph), and `replay()` re-launches the captured kernels with no host
work in between.  Inputs are STATIC: copy the next batch into the tensors the callable closed over (`GraphedStep.copy_inputs`), then
replay.  Results (the loss tensor the callable returned) are static tensors too.

Limits, by construction of graph capture: nothing inside the callable may synchronise or read device values on the host (the
mining strategies of loss/triplet_loss.py that build index lists on the host cannot be captured; 'noise_contrastive' and the
memory-bank step can), shapes must not change (a ragged last batch runs eagerly), and DistributedDataParallel's reducer is left to
eager steps.  The replayed step is bit-identical to the eager one (tests/test_train_loop_gpu.py::test_graphed_step_equals_eager).
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3, static_inputs=()):
        """fn(): one whole step on static tensors, returns a tensor (or tuple of tensors) to keep — e.g. the loss"""
        self.fn = fn
        self.static_inputs = tuple(static_inputs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def copy_inputs(self, *new):
        assert len(new) == len(self.static_inputs)
        for dst, src in zip(self.static_inputs, new):
            dst.copy_(src, non_blocking=True)

    def replay(self):
        self.graph.replay()
        return self.out

    __call__ = replay
