#!/usr/bin/env python3
"""HIP-graph capture (CLIP tower, decode step) inside a process that owns a live RCCL communicator (watchdog thread
polling events), as every rank of `bench.py --gpus N` does.  World size 1 is enough to have the communicator."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interactvlm_amd import model as M, synth, synthetic, weights as Wt  # noqa: E402
from interactvlm_amd.dist import gather_contacts  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    x = torch.ones(4, device=dev)
    dist.all_reduce(x)
    cfg = synthetic.config_tiny()
    cfg.llama = Wt.LlamaCfg(hidden=512, layers=2, heads=4, inter=1024, vocab=32003)  # a size the fused decode paths accept
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=6)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    outs = []
    for it in range(3):
        o = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)
        outs.append(gather_contacts(o["pred_contact_3d"]).cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert m.llm._dgraph is not None, "decode graph was not used"
    assert len(m.model.visual_model.image_encoder._graphs) == 1 and len(m.model.visual_model.mask_decoder._graphs) >= 1
    B = 3  # batched path: packed prefill + batched decode graph, next to the communicator as well
    icb, imb = synthetic.images(cfg, dev, seed=2, batch=B)
    ob = m.evaluate_batch(icb, imb, [ids[0]] * B, [cams[0]] * B, [(1024, 1024)] * B, [(1024, 1024)] * B, forced_new_tokens=forced)
    allc = gather_contacts(torch.cat([o["pred_contact_3d"] for o in ob])).cpu()
    assert allc.shape == (B, 6890)
    dist.barrier()
    dist.destroy_process_group()
    print("graphs + RCCL communicator: ok", tuple(outs[0].shape))


if __name__ == "__main__":
    main()
