"""Host mirror of the reference's inference engine core (crane-serve/src/engine/{mod,scheduler,sequence,types}.rs)
over the C ABI's cm_engine_* entry points: FIFO prefill-priority continuous batching on the paged KV pool.
Tokenizer / HTTP stay with the caller; this speaks token ids."""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib

FINISH_REASONS = {0: None, 1: "stop", 2: "length", 3: "cancelled"}      # Sequence::finish_reason (sequence.rs:117-125)


@dataclass
class GenerationParams:
    """engine/types.rs:26-50 (server defaults: handlers/openai.rs:104-109)."""
    max_tokens: int = 256
    temperature: Optional[float] = 0.8
    top_p: Optional[float] = 0.95
    top_k: Optional[int] = 40
    repetition_penalty: float = 1.05
    frequency_penalty: float = 0.0
    presence_penalty: float = 0.0
    eos_token_id: Sequence[int] = ()
    seed: int = 0

    @classmethod
    def greedy(cls, max_tokens: int, eos_token_id: Sequence[int] = ()) -> "GenerationParams":
        return cls(max_tokens=max_tokens, temperature=0.0, top_p=None, top_k=None, repetition_penalty=1.0,
                   eos_token_id=eos_token_id)


@dataclass
class Event:
    req_id: int
    kind: str                       # "token" | "finished" | "error"
    token: int = 0
    finish_reason: Optional[str] = None
    prompt_tokens: int = 0
    completion_tokens: int = 0
    error: int = 0


class InferenceEngine:
    """InferenceEngine::{accept_request, run-loop body} (engine/mod.rs:169-271, 525-599, 622-641)."""

    def __init__(self, model, max_running: int = 0, repeat_last_n: int = 64, seed: int = 0, batch_prefill: bool = True):
        self._lib = _lib.load()
        self._model = model                       # keeps the model alive
        o = _lib.CmEngineOpts()
        o.max_running, o.repeat_last_n, o.seed = max_running, repeat_last_n, seed
        o.batch_prefill = 0 if batch_prefill else -1       # several waiting prompts per pass over the weights (cm_prefill_batch)
        h = C.c_void_p()
        rc = self._lib.cm_engine_create(model._h, C.byref(o), C.byref(h))
        if rc != 0:
            raise _lib.CraneError(rc, "cm_engine_create failed")
        self._h = h
        self._ev = (_lib.CmEngineEvent * 2048)()          # room for 8 scheduler steps of 128 running sequences per native call

    def close(self):
        if self._h:
            self._lib.cm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        raise _lib.CraneError(rc, self._lib.cm_engine_last_error(self._h).decode())

    def submit(self, tokens: Sequence[int], params: GenerationParams) -> int:
        arr = np.ascontiguousarray(np.asarray(tokens, dtype=np.uint32))
        r = _lib.CmRequest()
        r.tokens = arr.ctypes.data_as(C.POINTER(C.c_uint32))
        r.n_tokens = arr.size
        r.max_tokens = params.max_tokens
        r.temperature = -1.0 if params.temperature is None else float(params.temperature)
        r.top_p = -1.0 if params.top_p is None else float(params.top_p)
        r.top_k = 0 if params.top_k is None else int(params.top_k)
        r.repetition_penalty = float(params.repetition_penalty)
        r.frequency_penalty = float(params.frequency_penalty)
        r.presence_penalty = float(params.presence_penalty)
        eos = list(params.eos_token_id)
        for i in range(4):
            r.eos_token_id[i] = int(eos[i]) if i < len(eos) else -1
        r.seed = params.seed
        rid = C.c_uint64(0)
        rc = self._lib.cm_engine_submit(self._h, C.byref(r), C.byref(rid))
        if rc != 0:
            self._err(rc)
        return int(rid.value)

    def cancel(self, req_id: int):
        rc = self._lib.cm_engine_cancel(self._h, req_id)
        if rc != 0:
            self._err(rc)

    def has_work(self) -> bool:
        return bool(self._lib.cm_engine_has_work(self._h))

    def step(self, max_steps: int = 1) -> List[Event]:
        """One scheduling decision (max_steps = 1, cm_engine_step) or up to max_steps of them in one native call."""
        n = C.c_size_t(0)
        if max_steps <= 1:
            rc = self._lib.cm_engine_step(self._h, self._ev, len(self._ev), C.byref(n))
        else:
            rc = self._lib.cm_engine_step_many(self._h, max_steps, self._ev, len(self._ev), C.byref(n))
        if rc != 0:
            self._err(rc)
        out = []
        for i in range(n.value):
            e = self._ev[i]
            kind = ("token", "finished", "error")[e.kind]
            out.append(Event(int(e.req_id), kind, int(e.token), FINISH_REASONS.get(e.finish_reason), int(e.prompt_tokens),
                             int(e.completion_tokens), int(e.error)))
        return out

    def stats(self) -> Dict[str, int]:
        s = _lib.CmEngineStats()
        self._lib.cm_engine_get_stats(self._h, C.byref(s))
        return {n: int(getattr(s, n)) for n, _ in s._fields_ if n != "reserved"}

    def run_until_idle(self, max_steps: int = 1 << 20, steps_per_call: int = 1):
        """Drive the loop; returns ({req_id: generated tokens}, {req_id: finished Event})."""
        toks: Dict[int, List[int]] = {}
        done: Dict[int, Event] = {}
        for _ in range(max_steps):
            if not self.has_work():
                break
            for e in self.step(steps_per_call):
                if e.kind == "token":
                    toks.setdefault(e.req_id, []).append(e.token)
                else:
                    done[e.req_id] = e
        return toks, done
