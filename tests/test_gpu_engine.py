"""GPU: the continuous-batching engine (engine.cpp) -- scheduling rules of crane-serve/src/engine/scheduler.rs:67-98,
stop rules of sequence.rs:75-125, preemption of engine/mod.rs:430-504 -- checked against single-request generate()."""
import numpy as np
import pytest

from crane_amd import configs

pytestmark = pytest.mark.gpu


def _model(name="tiny-qwen3", **kw):
    from crane_amd.backend import Model
    kw.setdefault("max_seq_len", 256)
    kw.setdefault("max_seqs", 8)
    kw.setdefault("kv_dtype", "f32")
    return Model.synthetic(configs.get_config(name), seed=0, **kw)


def _prompts(V, n):
    return [[(7 * i + 3 + 11 * j) % V for i in range(5 + 3 * j)] for j in range(n)]


def _reference_greedy(m, prompt, n, eos=()):
    from crane_amd.backend import GenerationConfig
    out = m.generate(prompt, GenerationConfig.greedy(n, eos_token_id=eos[0] if eos else None))
    return out[len(prompt):]


@pytest.mark.parametrize("name", ["tiny-qwen3", "tiny-qwen3.5"])
def test_engine_greedy_matches_single_request_generate(name):
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model(name)
    try:
        prompts = _prompts(m.vocab_size, 5)
        lens = [6, 11, 3, 9, 14]
        want = [_reference_greedy(m, p, n) for p, n in zip(prompts, lens)]
        m.clear_kv_cache()
        eng = InferenceEngine(m, max_running=4)
        ids = [eng.submit(p, GenerationParams.greedy(n)) for p, n in zip(prompts, lens)]
        toks, done = eng.run_until_idle()
        for rid, w, p in zip(ids, want, prompts):
            assert toks[rid] == w
            assert done[rid].kind == "finished" and done[rid].finish_reason == "length"
            assert done[rid].prompt_tokens == len(p) and done[rid].completion_tokens == len(w)
        st = eng.stats()
        assert st["completed"] == 5 and st["failed"] == 0 and st["waiting"] == 0 and st["running"] == 0
        assert st["prefill_steps"] == 5 and st["free_pages"] == st["total_pages"]
        assert st["completion_tokens"] == sum(lens)
        eng.close()
    finally:
        m.close()


def test_schedule_is_prefill_priority_then_batched_decode():
    """the reference's schedule, step by step: ONE prompt per prefill step (scheduler.rs:67-98) -- batch_prefill off"""
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model()
    try:
        eng = InferenceEngine(m, max_running=2, batch_prefill=False)
        prompts = _prompts(m.vocab_size, 3)
        a, b, c = (eng.submit(p, GenerationParams.greedy(n)) for p, n in zip(prompts, [3, 5, 2]))
        ev = eng.step(); assert [(e.req_id, e.kind) for e in ev] == [(a, "token")]          # prefill a
        ev = eng.step(); assert [(e.req_id, e.kind) for e in ev] == [(b, "token")]          # prefill b (running 1 < 2)
        ev = eng.step(); assert [(e.req_id, e.kind) for e in ev] == [(a, "token"), (b, "token")]   # decode round, c waits
        ev = eng.step()                                                                       # a reaches 3 tokens
        assert [(e.req_id, e.kind) for e in ev] == [(a, "token"), (b, "token"), (a, "finished")]
        ev = eng.step(); assert [(e.req_id, e.kind) for e in ev] == [(c, "token")]          # slot free: prefill c
        ev = eng.step()
        assert [(e.req_id, e.kind) for e in ev] == [(b, "token"), (c, "token"), (c, "finished")]
        toks, done = eng.run_until_idle()
        assert done[b].completion_tokens == 5 and not eng.has_work()
        eng.close()
    finally:
        m.close()


def test_stop_rules_eos_and_zero_max_tokens():
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model()
    try:
        p = _prompts(m.vocab_size, 1)[0]
        free = _reference_greedy(m, p, 8)
        eos = free[3]
        first = free.index(eos)
        m.clear_kv_cache()
        eng = InferenceEngine(m)
        r1 = eng.submit(p, GenerationParams.greedy(8, eos_token_id=[999999 % m.vocab_size, eos]))
        r2 = eng.submit(p, GenerationParams.greedy(0))          # prefill still samples one token (engine/mod.rs:716-731)
        toks, done = eng.run_until_idle()
        assert toks[r1] == free[:first + 1] and done[r1].finish_reason == "stop"            # EOS is emitted, then stop
        assert toks[r2] == free[:1] and done[r2].finish_reason == "length" and done[r2].completion_tokens == 1
        eng.close()
    finally:
        m.close()


def test_submit_rejections_and_cancel():
    from crane_amd._lib import CraneError
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model()
    try:
        eng = InferenceEngine(m, batch_prefill=False)          # (the step counts below are the one-prompt-per-step schedule's)
        with pytest.raises(CraneError):
            eng.submit([], GenerationParams.greedy(4))
        with pytest.raises(CraneError, match="exceeds server max_seq_len"):
            eng.submit([1] * 300, GenerationParams.greedy(4))
        with pytest.raises(CraneError):
            eng.submit([m.vocab_size], GenerationParams.greedy(4))
        assert eng.stats()["failed"] == 3
        # max_tokens is clamped to max_seq_len - prompt_len (effective_max_tokens, engine/mod.rs:507-513)
        long = eng.submit([1] * 250, GenerationParams.greedy(1000))
        other = eng.submit([2, 3, 4], GenerationParams.greedy(50))
        eng.step(); eng.step(); eng.step()
        eng.cancel(other)
        toks, done = eng.run_until_idle()
        assert done[other].finish_reason == "cancelled" and done[other].completion_tokens == 2 and other not in toks
        assert done[long].completion_tokens == 6 and done[long].finish_reason == "length" and len(toks[long]) == 4
        assert eng.stats()["free_pages"] == eng.stats()["total_pages"]
        eng.close()
    finally:
        m.close()


def test_preemption_when_the_page_pool_runs_out():
    from crane_amd.engine import GenerationParams, InferenceEngine
    # 6 pages of 64 tokens: three sequences growing past 128 tokens cannot all stay resident
    m = _model(kv_pool_tokens=6 * 64, max_seqs=4)
    try:
        prompts = [[(5 * i + j) % m.vocab_size for i in range(60)] for j in range(3)]
        want = [_reference_greedy(m, p, 80) for p in prompts]
        m.clear_kv_cache()
        eng = InferenceEngine(m, max_running=3)
        ids = [eng.submit(p, GenerationParams.greedy(80)) for p in prompts]
        toks, done = eng.run_until_idle()
        st = eng.stats()
        assert st["preemptions"] >= 1 and st["completed"] == 3 and st["failed"] == 0
        assert st["free_pages"] == st["total_pages"]
        for rid, w in zip(ids, want):
            assert toks[rid] == w and done[rid].finish_reason == "length"
        eng.close()
    finally:
        m.close()


def test_sampled_requests_are_seeded_and_bounded():
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model()
    try:
        prompts = _prompts(m.vocab_size, 4)

        def run(seed):
            eng = InferenceEngine(m, seed=seed)
            ids = [eng.submit(p, GenerationParams(max_tokens=12, temperature=30.0, top_p=0.95, top_k=40, repetition_penalty=1.05))
                   for p in prompts]
            toks, done = eng.run_until_idle()
            eng.close()
            return [toks[i] for i in ids]

        a, b, c = run(1), run(1), run(2)
        assert a == b and a != c
        assert all(0 <= t < m.vocab_size for seq in a for t in seq) and all(len(s) == 12 for s in a)
        # top_k = 1 with any temperature == greedy; penalties off
        eng = InferenceEngine(m)
        rid = eng.submit(prompts[0], GenerationParams(max_tokens=10, temperature=0.7, top_p=None, top_k=1, repetition_penalty=1.0))
        toks, _ = eng.run_until_idle()
        eng.close()
        assert toks[rid] == _reference_greedy(m, prompts[0], 10)
    finally:
        m.close()


def test_row_sampler_equals_the_per_row_sampler():
    """Model::sample_enqueue_rows (every sampled row of a decode group / prompt pass in one set of launches, blockIdx.y = row)
    runs the per-row sampler's device code on per-slot scratch: the same tokens as cm_debug_set("sample_rows", 0) -- mixed
    rows: top-k + top-p with penalties, top-k only, greedy with a repetition penalty, plain greedy, and a full-vocabulary
    Gumbel row (no top-k / top-p: stays on the per-row path inside the same group)."""
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model(max_seqs=16)
    try:
        prompts = _prompts(m.vocab_size, 20)
        params = [GenerationParams(max_tokens=14, temperature=0.9, top_p=0.95, top_k=40, repetition_penalty=1.05),
                  GenerationParams(max_tokens=14, temperature=1.3, top_p=None, top_k=8, repetition_penalty=1.0),
                  GenerationParams(max_tokens=14, temperature=0.0, top_p=None, top_k=0, repetition_penalty=1.3),
                  GenerationParams.greedy(14),
                  GenerationParams(max_tokens=14, temperature=1.1, top_p=None, top_k=0, repetition_penalty=1.0, frequency_penalty=0.2)]

        def run(rows):
            m.debug_set("sample_rows", rows)
            eng = InferenceEngine(m, seed=3)
            ids = [eng.submit(p, params[i % len(params)]) for i, p in enumerate(prompts)]
            toks, done = eng.run_until_idle()
            eng.close()
            return [toks[i] for i in ids]

        a, b = run(1), run(0)
        assert a == b
        assert all(len(t) == 14 for t in a)
    finally:
        m.debug_set("sample_rows", 1)
        m.close()


def test_step_many_matches_single_steps():
    """cm_engine_step_many: several scheduling decisions per native call, same event stream."""
    from crane_amd import configs
    from crane_amd.backend import Model
    from crane_amd.engine import GenerationParams, InferenceEngine
    cfg = configs.get_config("tiny-qwen3")
    V = cfg["vocab_size"]
    outs = []
    for spc in (1, 5):
        m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=8)
        try:
            eng = InferenceEngine(m, max_running=3)
            ids = [eng.submit([(7 * k + 3 + 13 * i) % V for k in range(6 + i)],
                              GenerationParams.greedy(9) if i % 2 else GenerationParams(max_tokens=9, seed=5)) for i in range(6)]
            toks, done = eng.run_until_idle(steps_per_call=spc)
            outs.append(([toks[i] for i in ids], [done[i].finish_reason for i in ids]))
            eng.close()
        finally:
            m.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("name", ["tiny-qwen3", "tiny-qwen3.5", "tiny-qwen3-untied"])
def test_batched_prompt_pass_changes_no_token(name):
    """cm_engine_opts.batch_prefill (default on): waiting prompts share ONE pass over the weights (cm_prefill_batch: GEMMs over all
    rows, RoPE / KV append / causal attention / GDN scan per sequence).  Same tokens, same finish events as the
    one-prompt-per-step schedule -- greedy and with the server-default sampler (per-request seeds) -- and the prompts of a step
    arrive in queue order."""
    from crane_amd.engine import GenerationParams, InferenceEngine
    m = _model(name, max_seqs=12)
    try:
        V = m.vocab_size
        prompts = [[(7 * i + 3 + 11 * j) % V for i in range(2 + (5 * j) % 23)] for j in range(10)]
        lens = [4 + (3 * j) % 7 for j in range(10)]
        outs = []
        for bp in (False, True):
            m.clear_kv_cache()
            eng = InferenceEngine(m, max_running=6, batch_prefill=bp, seed=7)
            ids = []
            for j, (p, n) in enumerate(zip(prompts, lens)):
                gp = GenerationParams.greedy(n) if j % 2 == 0 else GenerationParams(max_tokens=n, temperature=0.8, top_p=0.95, top_k=40, repetition_penalty=1.05)
                ids.append(eng.submit(p, gp))
            first = []
            if bp:
                first = eng.step()                  # 6 slots: the first six prompts in one pass, in queue order
                assert [(e.req_id, e.kind) for e in first if e.kind == "token"] == [(i, "token") for i in ids[:6]]
            toks, done = eng.run_until_idle()
            for e in reversed(first):               # (the events of the step taken by hand belong in front)
                if e.kind == "token":
                    toks.setdefault(e.req_id, []).insert(0, e.token)
                else:
                    done[e.req_id] = e
            outs.append(([toks[i] for i in ids], [(done[i].finish_reason, done[i].completion_tokens) for i in ids]))
            assert eng.stats()["prefill_steps"] == 10 and eng.stats()["free_pages"] == eng.stats()["total_pages"]
            eng.close()
        assert outs[0] == outs[1]
    finally:
        m.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "tiny-qwen3.5"])
def test_prefill_batch_equals_per_sequence_prefill(name):
    """cm_prefill_batch against cm_seq_forward on the same prompts: last-position logits (the rows only meet other rows inside
    GEMM tiles: summation order of a split-K at most), greedy ids, and the decode step that follows (pages / GDN state written
    by the batched pass)."""
    m = _model(name, max_seqs=12, max_seq_len=512)
    try:
        V = m.vocab_size
        prompts = [[(13 * j + 7 * k + 3) % V for k in range(n)] for j, n in enumerate([1, 5, 64, 65, 130, 2, 33])]
        ref = []
        for p in prompts:
            s_ = m.seq_alloc()
            lg, g = m.seq_forward(s_, p, 0)
            nxt, _ = m.seq_forward(s_, [int(g)], len(p))
            ref.append((lg.copy(), int(g), nxt.copy()))
            m.seq_free(s_)
        seqs = [m.seq_alloc() for _ in prompts]
        lg, gr = m.prefill_batch(seqs, prompts)
        for i, p in enumerate(prompts):
            assert float(np.abs(lg[i] - ref[i][0]).max() / np.abs(ref[i][0]).max()) < 2e-5, i
            assert int(gr[i]) == ref[i][1]
            assert m.seq_len(seqs[i]) == len(p)
        lg2, _ = m.step_batch_decode(seqs, [int(t) for t in gr])
        for i in range(len(prompts)):
            assert float(np.abs(lg2[i, 0] - ref[i][2]).max() / np.abs(ref[i][2]).max()) < 2e-5, i
    finally:
        m.close()
