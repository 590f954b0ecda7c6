import sys, torch
sys.path.insert(0, "/root/repo")
from interactvlm_amd import llava, ops, _lib, weights as Wt
from interactvlm_amd.weights import synth_weights
dev = torch.device("cuda:0")
hidden, heads, inter, vocab = 1024, 8, 1024, 256
lc = Wt.LlamaCfg(hidden=hidden, layers=2, heads=heads, inter=inter, vocab=vocab)
w = {k: v.to(torch.bfloat16).float() for k, v in synth_weights(Wt.llama_spec(lc)).items()}
llm = llava.Llama(w, lc, dev, max_len=64)
T0 = 20
lib = _lib.load()
nbytes = lib.ivlm_llama_generate_workspace_bytes(hidden, inter)
k = torch.arange(hidden, device=dev)
llm.lm_head.copy_(((k % 128).float() + 128 * (k // 128).float() / 8)[None].expand(vocab, hidden).to(torch.bfloat16))
def run(x):
    hid = torch.zeros(T0 + 4, hidden, dtype=torch.bfloat16, device=dev); hid[T0 - 1] = x
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    new_ids = torch.zeros(4, dtype=torch.int32, device=dev); arg = torch.zeros(4, dtype=torch.int32, device=dev)
    c = lc
    rc = lib.ivlm_llama_generate(llm.layer_ptrs.data_ptr(), c.layers, heads, hidden // heads, hidden, inter, vocab, c.eps,
        (hidden // heads) ** -0.5, llm.rope[0].data_ptr(), llm.rope[1].data_ptr(), llm.kcache.data_ptr(), llm.vcache.data_ptr(),
        llm.kcache.stride(0), 64, llm.embed.data_ptr(), llm.norm.data_ptr(), llm.lm_head.data_ptr(), hid.data_ptr(), T0, 1, -1,
        0, new_ids.data_ptr(), arg.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cv = ws[4096:4096 + 1024].view(torch.float32)
    return cv[:4].tolist(), (x.float() @ llm.lm_head.float().T)[:2].tolist()
for e in [0, 1, 2, 7, 8, 9, 15, 16, 63, 64, 127, 128, 511, 512, 513, 1023]:
    x = torch.zeros(hidden, dtype=torch.bfloat16, device=dev); x[e] = 1
    print("onehot", e, run(x))
x = torch.ones(hidden, dtype=torch.bfloat16, device=dev)
print("ones", run(x))
