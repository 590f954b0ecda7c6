"""Stage times of one evaluate() on the main stream (HIP events, no profiler): CLIP + prefill, decode loop, tail - with the SAM
encoder running concurrently on the side stream and without it (embeddings given).  Shows where the two-stream overlap costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from interactvlm_amd import model as M
    from interactvlm_amd import synthetic

    dev = torch.device("cuda:0")
    if os.environ.get("IVLM_TILE"):  # force the GEMM block tile (64 / 96 / 128 / 256 / 512)
        from interactvlm_amd import _lib
        _lib.load().ivlm_gemm_tile_override(int(os.environ["IVLM_TILE"]))
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    vid, bary = synthetic.body_lift_tables(dev)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=(vid, bary))
    del w
    if os.environ.get("IVLM_NO_FUSE_ATTN_OPROJ"):
        m.llm.fuse_attn_oproj = False
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    emb_cached = m.precompute_visual_embs(im[0])
    main_s = torch.cuda.current_stream(dev)
    side = m._side_stream

    def one(with_sam):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        sam_done = torch.cuda.Event(enable_timing=True)
        box = {}
        ev[0].record(main_s)
        if with_sam:
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                box["emb"] = m.model.visual_model.image_encoder(im[0])
                sam_done.record(side)
        else:
            box["emb"] = emb_cached
        ev[1].record(main_s)  # host done launching SAM
        out_ids, hidden = m.generate(ic, ids, 32, 2, forced, after_prefill=lambda: ev[2].record(main_s))
        ev[3].record(main_s)
        if with_sam:
            main_s.wait_event(sam_done)
        rows = m._seg_rows(out_ids[0], extra_false_col=False)
        pm, _ = m._decode_sample(hidden, rows, out_ids[0], cams[0], box["emb"], (S, S), (S, S))
        pc = m.human_3d_contact_predictor([pm])
        pc.cpu()
        ev[4].record(main_s)
        torch.cuda.synchronize()
        t = [ev[0].elapsed_time(e) for e in ev[1:]]
        sam_t = ev[0].elapsed_time(sam_done) if with_sam else 0.0
        return t, sam_t

    for with_sam in (False, True, False, True):
        for _ in range(2):
            one(with_sam)
        acc = [0.0] * 4
        sam_acc = 0.0
        n = 5
        for _ in range(n):
            t, st = one(with_sam)
            acc = [a + x for a, x in zip(acc, t)]
            sam_acc += st
        t = [a / n for a in acc]
        print(f"SAM concurrent={with_sam}: host-launched SAM at {t[0]:.2f} ms | prefill done {t[1]:.2f} | decode done {t[2]:.2f} "
              f"(decode {t[2] - t[1]:.2f}) | end {t[3]:.2f} (tail {t[3] - t[2]:.2f}) | SAM encoder done at {sam_acc / n:.2f} ms", flush=True)


def schedules():
    """End-to-end evaluate() under the two launch orders of the SAM encoder: at t = 0 (default) or after the prefill."""
    import time

    from interactvlm_amd import model as M
    from interactvlm_amd import synthetic

    dev = torch.device("cuda:0")
    cfg = synthetic.config_7b()
    w = synthetic.device_weights(cfg, dev, seed=0)
    vid, bary = synthetic.body_lift_tables(dev)
    m = M.InteractVLMForCausalLM(cfg, w, dev, lift_tables=(vid, bary))
    del w
    ids, forced = synthetic.prompt_ids(cfg)
    cams = synthetic.human_cam_params()
    ic, im = synthetic.images(cfg, dev)
    S = cfg.sam.img_size
    for mode in ("t0", "after_prefill", "t0", "after_prefill"):
        m.sam_after_prefill = mode == "after_prefill"
        for _ in range(2):
            m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].cpu()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 6
        for _ in range(n):
            m.evaluate(ic, im, ids, cams, [(S, S)], [(S, S)], forced_new_tokens=forced)["pred_contact_3d"].cpu()
        torch.cuda.synchronize()
        print(f"SAM encoder launched {mode}: {1e3 * (time.perf_counter() - t) / n:.2f} ms per evaluate", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "schedules":
        schedules()
    else:
        main()
