"""Worker of test_data_parallel_training_driver: runs the stage-2 train() under torchrun (2 replicas on the one GPU of the
test box, gloo backend) and saves each rank's final parameters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from pronerf_amd import run_S_eS_eN_alter_base_refine2 as s2
    cfg, out_dir = sys.argv[1], sys.argv[2]
    tr, log = s2.train(['--config', cfg, '--max_steps', '6'], device='cuda:0')
    np.savez(os.path.join(out_dir, f'rank{os.environ["RANK"]}.npz'), param=tr.flat('param').cpu().numpy(), loss=np.array([e[1] for e in log]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
