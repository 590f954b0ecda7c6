"""smoke(): one tiny end-to-end evaluate() of the facade on cuda:0, checked against the CPU oracle."""
from __future__ import annotations

import torch


def run(dev):
    from interactvlm_amd import model as M
    from interactvlm_amd import synth, synthetic
    from interactvlm_amd import weights as Wt
    from oracle import pipeline as P

    cfg = synthetic.config_tiny()
    w = Wt.synth_weights(Wt.ivlm_spec(cfg))
    w = {k: v.to(torch.bfloat16).float() for k, v in w.items()}  # bf16-representable, shared with the oracle
    tables = synth.synth_mesh_tables(4, 1024, 1024, 6890, fg=0.4, seed=0, patch=8)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=tables)
    ids, forced = synthetic.prompt_ids(cfg, n_prompt=40, n_answer=8)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    out = m.evaluate(ic, im, ids, cams, [(1024, 1024)], [(1024, 1024)], forced_new_tokens=forced)
    full_ids = torch.cat([ids[0], torch.tensor(forced)])
    ref = P.model_forward(w, cfg, im[0].float().cpu(), ic.float().cpu(), full_ids, cams[0], tables)
    got = out["pred_contact_3d"].float().cpu()
    err = float((got - ref["pred_contact"]).abs().max())
    merr = float((out["pred_masks"][0].cpu() - ref["pred_masks"]).abs().max())
    print(f"[smoke] tiny evaluate(): max|dp_contact| = {err:.2e}, max|dmask| = {merr:.3f}")
    assert err < 1e-2 and merr < 0.3, (err, merr)
