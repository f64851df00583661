import sys, time, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import bench, dalm_oracle as O
print("cores", os.cpu_count(), flush=True)
for nt in (16, 32, 64):
    torch.set_num_threads(nt)
    dev = torch.device("cpu")
    t0 = time.time()
    model = bench.build_models(dev, torch.float32, bert_layers=1, llama_layers=1); model.train()
    tb = time.time() - t0
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    batch = bench.synthetic_batch(dev, 0)
    def step():
        qh = model.retriever_model(batch["retriever_query_input_ids"], batch["retriever_query_attention_mask"])[0]
        ph = model.retriever_model(batch["retriever_passage_input_ids"], batch["retriever_passage_attention_mask"])[0]
        q = O.ref_retrieval_embed(qh, batch["retriever_query_attention_mask"]); p = O.ref_retrieval_embed(ph, batch["retriever_passage_attention_mask"])
        t1 = time.time()
        logits = model.generator_model(input_ids=batch["generator_input_input_ids"], attention_mask=batch["generator_input_attention_mask"]).logits
        t2 = time.time()
        out = O.ref_step_loss(q, p, logits, batch["generator_input_input_ids"], batch["generator_input_attention_mask"], batch["query_passage_input_len"], 100)
        t3 = time.time()
        out["loss"].backward(); opt.step(); model.zero_grad()
        t4 = time.time()
        return t1, t2, t3, t4
    t0 = time.time(); step(); w = time.time() - t0
    t0 = time.time(); t1, t2, t3, t4 = step(); tot = time.time() - t0
    print(f"threads {nt}: build {tb:.1f}s warm {w:.2f}s step {tot:.2f}s  [retr {t1-t0:.2f} gen {t2-t1:.2f} loss {t3-t2:.2f} bwd+opt {t4-t3:.2f}]", flush=True)
    del model, opt
