"""The torch.distributed leg of the multi-GPU path with the pieces that CAN run on one GPU: torch wraps the library's
exchange buffers without a copy (one HIP runtime in the process), `all_gather_into_tensor` accepts them in
place with backend nccl (= RCCL) at world_size 1, and a ShardedStepper-driven run equals snn_run."""
import os
import socket

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_sharded_stepper_over_rccl_world_size_1(snn):
    import torch
    import torch.distributed as dist
    from snn_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")        # single node: no interface discovery
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        lay = parity.Layout([(0, 10, 10)])
        net = parity.make_oracle(lay)
        net["gap_conductance"] = 10.0
        net["current_voltage"] = ob.uniform_array(1, 100, -65.0, 30.0)
        net.fill_graph(2, 0.5, 1.5)
        net["do_plasticity"] = 1
        dn = parity.device_from_oracle(snn, net, shard=(0, 1))
        plan = dn.exchange_plan()
        send, recv = parallel.exchange_tensors(plan, torch.device("cuda", 0))
        assert recv.data_ptr() == plan["recv"] and send.data_ptr() == plan["send"]      # views, not copies
        stride = plan["shard_stride"]
        assert plan["mode"] == "allgather" and plan["plane_id"] == [0]                 # electrical only: voltage + bits
        assert send.numel() == recv.numel() == stride + stride // 32                    # 4 B + 1 bit per neuron
        # the collective itself, in place, on memory owned by libsnn_amd.so
        dist.all_gather_into_tensor(recv, send)
        torch.cuda.synchronize()
        v = dn.get_attr(0, "current_voltage")
        assert np.array_equal(v.view(np.uint32), net["current_voltage"].view(np.uint32))
        # stream-ordered, as bench.py --gpus N does it: no host synchronisation inside the loop
        side = torch.cuda.Stream()
        with pytest.raises(ValueError):
            dn.set_stream(0)                      # the legacy default stream cannot be adopted
        dn.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            for _ in range(200):
                dn.step_begin()
                dist.all_gather_into_tensor(recv, send)
                dn.step_end()
        dn.synchronize()
        # the overlapped schedule of ShardedStepper.run (step_begin_local before the previous gather is waited for)
        stepper = parallel.ShardedStepper(dn, 0, 1, always_exchange=True, stream=side, device=torch.device("cuda", 0))
        dn.set_plasticity(0, do_plasticity=False)
        net["do_plasticity"] = 0
        net.run(200)
        stepper.run(150)
        dn.synchronize()
        net.run(150)
        parity.assert_state_equal(net, parity.pull_state(dn, net))
        parity.assert_graph_equal(net, dn)
        dn.close()
    finally:
        dist.destroy_process_group()
