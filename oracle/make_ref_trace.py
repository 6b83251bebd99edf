"""TorchScript traces of the REFERENCE'S OWN module (imported from /root/reference), so that bench.py's `cpu_baseline` can time the reference
itself on the GPU box's host cores, where /root/reference does not exist.  MEASUREMENT FIXTURE (git-ignored oracle/_ref/, travels with the
snapshot like the built libraries); usage:  python -m oracle.make_ref_trace

  oracle/_ref/reference_module_eval.pt    torch.jit.trace of models.MobileNetSkipAdd((224, 224), pretrained=False) in .eval()  (seed 0)
  oracle/_ref/reference_module_train.pt   the same module traced in .train() (batch-statistics BatchNorm; parameters are the trace's own, so
                                          L1Loss + torch.optim.SGD run on it exactly as on the eager module)
  oracle/_ref/reference_module_b1.pt      marker + self-check record: outputs of eager vs traced at batch 1 / 8 compared bit for bit

A trace records the ATen operator sequence the reference's nn.Module tree executes (conv2d, batch_norm, hardtanh, relu, upsample_nearest2d,
add): replaying it dispatches to the very same CPU kernels with the reference's own graph, and it is batch-size agnostic (checked below).
Nothing of the reference's SOURCE is stored, only the operator graph and seeded random weights.
"""
import os
import sys

import torch

from oracle import make_golden

OUT = os.path.join(make_golden.REPO, "oracle", "_ref")


def main():
    sys.dont_write_bytecode = True                     # do not litter /root/reference with __pycache__
    ref_models, _ = make_golden.import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    m = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
    g = torch.Generator().manual_seed(5)
    x1, x8 = torch.rand(1, 3, 224, 224, generator=g), torch.rand(8, 3, 224, 224, generator=g)
    m.eval()
    with torch.no_grad():
        ev = torch.jit.trace(m, x1, check_trace=False)
        same = bool(torch.equal(ev(x1), m(x1))) and bool(torch.equal(ev(x8), m(x8)))      # traced at batch 1, replayed at batch 8
    m.train()
    tr = torch.jit.trace(m, x8, check_trace=False)
    m.eval()
    ev.save(os.path.join(OUT, "reference_module_eval.pt"))
    tr.save(os.path.join(OUT, "reference_module_train.pt"))
    torch.save({"eager_equals_traced_b1_b8": same, "torch": torch.__version__, "class": type(m).__module__ + "." + type(m).__name__,
                "source": os.path.abspath(ref_models.__file__)}, os.path.join(OUT, "reference_module_b1.pt"))
    print("traced the reference module (eager == traced at batch 1 and 8: %s) -> %s" % (same, OUT))
    assert same


if __name__ == "__main__":
    main()
