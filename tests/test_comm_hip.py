"""GPU: the C-ABI's RCCL seam (csrc/comm.hip) on one rank — library binding
(the RCCL instance PyTorch bundles), unique id, communicator, and the two
collectives enqueued on torch's current stream.  Multi-rank behaviour is RCCL's;
what is checked here is everything the engine adds around it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_round_trip():
    from xrdslam_amd.engine import dist as xd
    comm = xd.RcclComm(torch.device('cuda:0'))
    assert comm.world == 1
    x = torch.arange(1 << 20, dtype=torch.float32, device='cuda:0')
    ref = x.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):        # on the stream the caller is on
        comm.all_reduce_sum(x)
    side.synchronize()
    assert torch.equal(x, ref)
    m = torch.tensor([3, -7, 11], dtype=torch.int32, device='cuda:0')
    comm.all_reduce_max_i32(m)
    torch.cuda.synchronize()
    assert m.tolist() == [3, -7, 11]
    comm.close()
