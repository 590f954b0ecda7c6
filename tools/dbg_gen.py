import sys, torch
sys.path.insert(0, "/root/repo")
from interactvlm_amd import llava, ops, _lib, weights as Wt
from interactvlm_amd.weights import synth_weights
dev = torch.device("cuda:0")
for hidden, heads, inter, vocab in [(512, 4, 1024, 1000), (1024, 8, 1376, 1003), (1024, 8, 1024, 256)]:
    lc = Wt.LlamaCfg(hidden=hidden, layers=2, heads=heads, inter=inter, vocab=vocab)
    w = {k: v.to(torch.bfloat16).float() for k, v in synth_weights(Wt.llama_spec(lc)).items()}
    llm = llava.Llama(w, lc, dev, max_len=64)
    T0 = 20
    g = torch.Generator().manual_seed(1)
    emb = (torch.randn(T0, hidden, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    hid = torch.zeros(T0 + 4, hidden, dtype=torch.bfloat16, device=dev)
    hid[:T0] = llm.forward(emb, 0)
    lib = _lib.load()
    nbytes = lib.ivlm_llama_generate_workspace_bytes(hidden, inter)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    new_ids = torch.zeros(4, dtype=torch.int32, device=dev); arg = torch.zeros(4, dtype=torch.int32, device=dev)
    c = lc
    rc = lib.ivlm_llama_generate(llm.layer_ptrs.data_ptr(), c.layers, heads, hidden // heads, hidden, inter, vocab, c.eps,
        (hidden // heads) ** -0.5, llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), llm.kcache.data_ptr(), llm.vcache.data_ptr(),
        llm.kcache.stride(0), 64, llm.embed.data_ptr(), llm.norm.data_ptr(), llm.lm_head.data_ptr(), hid.data_ptr(), T0, 1, -1,
        0, new_ids.data_ptr(), arg.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lg = hid[T0 - 1].float() @ llm.lm_head.float().T
    cv = ws[4096:4096 + 1024].view(torch.float32); ci = ws[8192:8192 + 1024].view(torch.int32)
    print("cfg", hidden, inter, vocab, "rc", rc, "status", ws[:8].view(torch.int32).tolist(), "arg", arg[0].item(), "ref", lg.argmax().item())
    # per-block check
    G = 256
    bad = 0
    for b in range(G):
        r0 = vocab * b // G; r1 = vocab * (b + 1) // G if b < G - 1 else vocab
        if r1 > r0:
            v, i = lg[r0:r1].max(0)
            if abs(v.item() - cv[b].item()) > 1e-2 * max(1, abs(v.item())) or i.item() + r0 != ci[b].item():
                bad += 1
                if bad < 6: print("  block", b, "rows", r0, r1, "ref", v.item(), i.item() + r0, "got", cv[b].item(), ci[b].item())
    print("  bad blocks", bad)
