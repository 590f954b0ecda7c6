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

    # [r6] finer events on the language stream (is the first language kernel late under the overlap?):
    # after the CLIP tower + projector, after LLaMA layers 0 / 7 / 15 / 23 of the prefill - no profiler attached.
    marks = {}
    orig_enc, orig_layer = m.encode_images, m.llm._layer_f16
    state = {"layer": 0, "after_clip": None}

    def enc(*a, **k):
        r = orig_enc(*a, **k)
        marks["clip"].record(main_s)
        if state["after_clip"] is not None:
            state["after_clip"]()
        return r

    def layer(*a, **k):
        r = orig_layer(*a, **k)
        i = state["layer"]
        if f"L{i}" in marks:
            marks[f"L{i}"].record(main_s)
        state["layer"] = i + 1
        return r

    m.encode_images, m.llm._layer_f16 = enc, layer
    names = ["clip", "L0", "L7", "L15", "L23"]

    def one(mode):
        """mode: None (embeddings given), 't0' (encoder enqueued first: the product), 'after_clip', 'after_prefill'"""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        sam_done = torch.cuda.Event(enable_timing=True)
        sam_start = torch.cuda.Event(enable_timing=True)
        for n in names:
            marks[n] = torch.cuda.Event(enable_timing=True)
        state["layer"] = 0
        box = {}

        def launch_sam():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                sam_start.record(side)
                box["emb"] = m.model.visual_model.image_encoder(im[0])
                sam_done.record(side)

        ev[0].record(main_s)
        state["after_clip"] = launch_sam if mode == "after_clip" else None
        if mode == "t0":
            launch_sam()
        elif mode is None:
            box["emb"] = emb_cached
        ev[1].record(main_s)  # host done launching SAM

        def after_prefill():
            ev[2].record(main_s)
            if mode == "after_prefill":
                launch_sam()

        out_ids, hidden = m.generate(ic, ids, 32, 2, forced, after_prefill=after_prefill)
        ev[3].record(main_s)
        if mode is not None:
            main_s.wait_event(sam_done)
        rows = m._seg_rows(out_ids[0], extra_false_col=False)
        pm, _ = m._decode_sample(hidden, rows, out_ids[0], cams[0], box["emb"], (S, S), (S, S))
        pc = m.human_3d_contact_predictor([pm])
        pc.cpu()
        ev[4].record(main_s)
        torch.cuda.synchronize()
        t = [ev[0].elapsed_time(e) for e in ev[1:]] + [ev[0].elapsed_time(marks[n]) for n in names]
        sam_t = (ev[0].elapsed_time(sam_start), ev[0].elapsed_time(sam_done)) if mode is not None else (0.0, 0.0)
        return t, sam_t

    order = (None, "t0", "after_clip", "after_prefill") * 2
    for mode in order:
        for _ in range(2):
            one(mode)
        n = 5
        acc = None
        sam_acc = [0.0, 0.0]
        for _ in range(n):
            t, st = one(mode)
            acc = t if acc is None else [a + x for a, x in zip(acc, t)]
            sam_acc = [a + x for a, x in zip(sam_acc, st)]
        t = [a / n for a in acc]
        print(f"SAM encoder {str(mode):13s}: marker after SAM launch {t[0]:.2f} ms | CLIP done {t[4]:.2f} | layer 0 / 7 / 15 / 23 done "
              f"{t[5]:.2f} / {t[6]:.2f} / {t[7]:.2f} / {t[8]:.2f} | prefill done {t[1]:.2f} | decode done {t[2]:.2f} (decode {t[2] - t[1]:.2f}) | "
              f"end {t[3]:.2f} (tail {t[3] - t[2]:.2f}) | SAM encoder {sam_acc[0] / n:.2f} .. {sam_acc[1] / n:.2f} ms", flush=True)


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
