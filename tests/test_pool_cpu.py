"""Host logic of the activation-buffer pool (iic_amd.ops.PTPool): ownership by the pool rather than
by address, idempotent release, the evaluation sweep, and the second-backward guard."""
import pytest
import torch

from iic_amd import ops


def test_pool_recycles_and_ignores_foreign_and_double_release():
  pool = ops.PTPool()
  a = pool.alloc((2, 6, 6, 8), "cpu", 1)
  assert a.dtype == ops.PT_DTYPE[0] and float(a.float().abs().sum()) == 0.0
  pool.release(a)
  pool.release(a)                                   # second release: ignored
  key = next(iter(pool.free))
  assert len(pool.free[key]) == 1
  b = pool.alloc((2, 6, 6, 8), "cpu", 1)
  assert b.data_ptr() == a.data_ptr()               # recycled
  assert pool.alloc((2, 6, 6, 8), "cpu", 1).data_ptr() != a.data_ptr()
  foreign = torch.zeros((2, 6, 6, 8), dtype=ops.PT_DTYPE[0])
  pool.release(foreign)                             # never one of ours
  assert all(t.data_ptr() != foreign.data_ptr() for l in pool.free.values() for t in l)
  pool.release(b[:1])                               # a view of an owned buffer is not the buffer
  assert b.data_ptr() in pool.live
  # a different border width is a different class of buffer
  pool.release(b)
  c = pool.alloc((2, 6, 6, 8), "cpu", 2)
  assert c.data_ptr() != b.data_ptr()


def test_pool_keeps_its_buffers_alive():
  pool = ops.PTPool()
  ptrs = set()
  for _ in range(4):
    t = pool.alloc((1, 4, 4, 8), "cpu", 1)
    ptrs.add(t.data_ptr())
    del t                                           # dropped without release (an evaluation forward)
  assert len(ptrs) == 4 and len(pool.owned) == 4    # no address came back from the allocator
  assert pool.allocated_bytes == 4 * 4 * 4 * 8 * 2


def test_sweep_returns_only_buffers_since_the_mark():
  pool = ops.PTPool()
  keep = pool.alloc((1, 4, 4, 8), "cpu", 1)         # a training forward's saved activation
  mark = pool.mark()
  e1 = pool.alloc((1, 4, 4, 8), "cpu", 1)
  e2 = pool.alloc((1, 4, 4, 8), "cpu", 1)
  pool.release(e1)                                  # released normally inside the forward
  pool.sweep(mark)
  assert keep.data_ptr() in pool.live and not {e1.data_ptr(), e2.data_ptr()} & set(pool.live)
  assert sorted(t.data_ptr() for t in pool.free[next(iter(pool.free))]) == sorted([e1.data_ptr(), e2.data_ptr()])


def test_auto_branch_wrapper_sweeps_under_no_grad(monkeypatch):
  pool = ops.PTPool()
  monkeypatch.setattr(ops, "POOL", pool)

  class M(torch.nn.Module):
    @ops.auto_branch
    def forward(self, x):
      buf = ops.pt_alloc(1, 2, 2, 8, 1, x.device)   # never released by the forward itself
      return x + float(buf.float().sum())

  m = M()
  with torch.no_grad():
    for _ in range(3):
      m(torch.ones(2))
  assert len(pool.owned) == 1 and not pool.live      # one buffer, recycled every call
  m(torch.ones(2))                                   # grad mode: the backward would release it
  assert len(pool.live) == 1


def test_second_backward_raises():
  class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
      ctx.branch = 0
      return x * 2

    @ops.branch_backward
    def backward(ctx, g):
      return g * 2

  x = torch.ones(3, requires_grad=True)
  y = F.apply(x).sum()
  y.backward(retain_graph=True)
  with pytest.raises(RuntimeError, match="second time"):
    y.backward()


def test_recycled_buffer_is_a_fresh_tensor_object():
  """Autograd state of a buffer's previous life (grad_fn, user hooks) must not leak into the next:
  Tensor.register_hook binds its hook dict to the grad_fn only the first time it is called on an
  object, so a recycled OBJECT would silently lose hooks registered in a later step."""
  pool = ops.PTPool()
  a = pool.alloc((1, 4, 4, 8), "cpu", 1)
  fired = []
  (a.float().requires_grad_(True) * 1.0).register_hook(lambda g: fired.append(1))
  a.requires_grad_(True)
  a.register_hook(lambda g: fired.append(2))
  pool.release(a)
  b = pool.alloc((1, 4, 4, 8), "cpu", 1)
  assert b.data_ptr() == a.data_ptr() and b is not a
  assert not b.requires_grad and b._backward_hooks is None and b.grad_fn is None
