"""Autoregressive generation for the MI355X engines: greedy, top-k sampling and beam search with a KV cache.

Replaces what the reference obtains from transformers' GenerationMixin (TF5 = transformers/generation/utils.py @ 5.15.0:
generate :2260, _sample :2783, _beam_search :3208) together with the reference's own per-step input assembly
`prepare_inputs_for_generation` (modules/transformers/longitudinal_model/modelling_longitudinal.py:251-295 and the
single/multi variants). Behaviour kept: BOS is prepended when no row of the prompt starts with it (:850-907) and the
longitudinal model strips it again before feeding the decoder (:270-271); finished rows keep emitting PAD (:2932);
`generate.__wrapped__` (no torch.no_grad) returns `scores` that carry autograd -- implemented as sample-then-recompute:
the sampled sequence is re-scored by ONE teacher-forced pass whose per-step inputs are exactly those of the cached steps.
"""
from __future__ import annotations

import torch

from . import ops


class DecodeSession:
    """Static state of one cached greedy / sampling decode (fixed batch, encoder length, maximum length) plus the hipGraphs of
    its single-token steps. A decode step is ~90 short kernels; launched eagerly it is bound by the host (~9 us per launch through
    Python/ctypes), so every step index is captured ONCE into a graph -- all buffers the kernels touch (token ids, masks, KV cache,
    logits, next-token word) are static, and the position-dependent scalars (cache row, attention length) are baked per step -- and
    replayed afterwards. Sessions are cached on the model and reused by every later generate() call of the same geometry (each SCST
    step decodes twice with identical shapes)."""

    def __init__(self, model, B, S, Lmax, has_mask, share=1):
        dev = model.device
        self.model, self.B, self.S, self.Lmax = model, B, S, Lmax
        D = model.config.decoder.hidden_size
        rows, B = B, B // share          # `share` decodes of the same studies (SCST: sample + greedy) read ONE copy of the encoder rows / cross K,V
        self.ids = torch.zeros((rows, Lmax), dtype=torch.int64, device=dev)
        self.unfinished = torch.ones(rows, dtype=torch.int32, device=dev)
        self.nxt = torch.zeros(rows, dtype=torch.int64, device=dev)
        self.enc16 = torch.empty((B, S, D), dtype=torch.bfloat16, device=dev)
        self.enc_mask8 = torch.empty((B, S), dtype=torch.uint8, device=dev) if has_mask else None
        self.enc_bits = torch.zeros((B, (S + 31) // 32), dtype=torch.int32, device=dev) if has_mask else None      # the same mask as bit words (decode kernels)
        self.cache = model._dec.new_cache(rows, Lmax, dev)
        L = model.config.decoder.num_hidden_layers
        if model._dec.cross_kv_fused():
            # cross-attention K / V of ALL layers as the column blocks of one [B, S, 2 L D] buffer: ONE projection GEMM at prefill, and the
            # teacher-forced re-scoring pass of an SCST step reads the same buffer instead of projecting the encoder output again
            self.cache.kv_all = torch.empty((B, S, 2 * L * D), dtype=torch.bfloat16, device=dev)
            self.cache.ck = [self.cache.kv_all[:, :, 2 * l * D:(2 * l + 1) * D] for l in range(L)]
            self.cache.cv = [self.cache.kv_all[:, :, (2 * l + 1) * D:(2 * l + 2) * D] for l in range(L)]
        else:
            self.cache.ck = [torch.empty((B, S, D), dtype=torch.bfloat16, device=dev) for _ in range(L)]
            self.cache.cv = [torch.empty((B, S, D), dtype=torch.bfloat16, device=dev) for _ in range(L)]
        if ops.attention_cross_mfma_ok(rows, B, S):             # static fragment-ordered K / V copies (captured by the step graphs)
            self.cache.cpk = [(torch.empty(B * S * D, dtype=torch.bfloat16, device=dev), torch.empty(B * S * D, dtype=torch.bfloat16, device=dev)) for _ in range(L)]
        self.seed = torch.zeros(1, dtype=torch.int32, device=dev)     # dropout seed of the running decode (train mode); graphs read it
        # uniforms of every sampling step of a decode, drawn by ONE torch call in reset(): a torch.rand inside a captured step costs three extra
        # tiny launches per replay (the generator's seed / offset refresh in front of the graph + the fill kernel itself)
        self.u_all = torch.zeros((Lmax, rows), dtype=torch.float32, device=dev)
        self.poll_host = torch.zeros(1, dtype=torch.int32).pin_memory()          # landing pad of the asynchronous EOS poll
        # per-step inputs of the cached steps (one fused kernel writes them): last token, token type, position, attention mask, and the
        # token-type / position history the teacher-forced re-scoring of the sampled rows needs
        self.new_id = torch.zeros((rows, 1), dtype=torch.int64, device=dev)
        self.tt1 = torch.zeros((rows, 1), dtype=torch.int64, device=dev)
        self.pos1 = torch.zeros((rows, 1), dtype=torch.int64, device=dev)
        self.mask8 = torch.zeros((rows, (Lmax + 7) // 8 * 8), dtype=torch.uint8, device=dev)[:, :Lmax]      # 8-byte aligned rows: one load per lane
        self.tt_hist = torch.zeros((rows, Lmax), dtype=torch.int64, device=dev)
        self.pos_hist = torch.zeros((rows, Lmax), dtype=torch.int64, device=dev)
        self.graphs = {}
        self.pool = None
        self.version = -1
        self.fills = 0                                     # number of reset()s = prefills: every one overwrites the cross K / V (decoder.SharedCrossKV)
        self.beam = None                                   # device-side beam-search state (beam_state)
        self._cache_home = (list(self.cache.k), list(self.cache.v), list(self.cache.k2), list(self.cache.v2))

    def beam_state(self, nb, pad):
        """State of a device-side beam search over this session's rows (BEAM-MAJOR: row = beam * studies + study, so the beams of a study
        share its cross-attention K/V): `self.ids` viewed [nb, B, Lmax] holds the running beams; finished beams, scores and the stop flags
        live here (csrc/decode.hip beam_step_kernel)."""
        dev, B = self.ids.device, self.B // nb
        if self.beam is None or self.beam["nb"] != nb:
            self.beam = dict(nb=nb, sequences=torch.zeros((nb, B, self.Lmax), dtype=torch.int64, device=dev),
                             run_scores=torch.zeros((B, nb), dtype=torch.float32, device=dev),
                             beam_scores=torch.zeros((B, nb), dtype=torch.float32, device=dev),
                             finished=torch.zeros((B, nb), dtype=torch.uint8, device=dev),
                             unsat=torch.zeros((2, B), dtype=torch.int32, device=dev), allhit=torch.zeros((2, B), dtype=torch.int32, device=dev),
                             beam_idx=torch.zeros(self.B, dtype=torch.int64, device=dev),
                             ws=ops.beam_ws(self.B, self.model.config.decoder.vocab_size, nb, dev))
        st = self.beam
        st["run_scores"].fill_(-1.0e9); st["run_scores"][:, 0] = 0.0            # only beam 0 of the identical start rows may continue
        st["beam_scores"].fill_(-1.0e9); st["finished"].zero_()
        st["unsat"].fill_(1); st["allhit"].zero_()
        st["sequences"].copy_(self.ids.view(nb, B, self.Lmax))
        return st

    def _swap_caches(self, n=1):
        c = self.cache
        if n % 2:
            c.k, c.k2 = c.k2, c.k
            c.v, c.v2 = c.v2, c.v

    def reset(self, ids0, enc16, enc_mask8):
        m = self.model
        m._dec.prepare()                           # weight-derived buffers (LoRA merge) are refreshed IN PLACE: captured pointers stay valid
        m._dec.refresh_decode_packs()              # ... as are the packed / LayerNorm-folded decode weights the replayed steps stream
        if self.version != m.flat16.data_ptr():    # parameters were re-packed (.to()/.cuda()): every captured pointer is stale
            self.graphs.clear()
            self.version = m.flat16.data_ptr()
        self.ids.fill_(0)
        self.ids[:, : ids0.shape[1]] = ids0
        self.unfinished.fill_(1)
        self.enc16.copy_(enc16)
        if self.enc_mask8 is not None:
            self.enc_mask8.copy_(enc_mask8)
            ops.pack_mask_bits(self.enc_mask8, out=self.enc_bits)
        self.cache.enc_bits = self.enc_bits
        self.cache.len = 0
        self.cache.cross_ready = False
        self.fills += 1
        self.cache.k, self.cache.v, self.cache.k2, self.cache.v2 = (list(t) for t in self._cache_home)      # beam search ping-pongs them
        if m.training:
            self.seed.copy_(m.next_dropout_seed())
        self.u_all.uniform_()                      # torch's CUDA generator: reproducible under torch.manual_seed

    def step(self, cur, strip, mode, n=1):
        """Append tokens cur .. cur+n-1 (0-based columns of self.ids). mode = (kind, special_token_ids, mask_token_id, top_k, temperature, eos, pad,
        train, top_p). The n cached steps are ONE hipGraph (a replay costs ~9 us of launch overhead whatever it holds: per token at n = 1, per
        8 tokens at n = 8)."""
        prefill = self.cache.len == 0
        if prefill or not self.model.graph_decode:
            for c in range(cur, cur + n):
                self._run(c, strip, mode, prefill and c == cur)
            return
        key = (cur, n, strip, mode)
        hit = self.graphs.get(key)
        if hit is None:
            g = torch.cuda.CUDAGraph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            keep = self.cache.len
            torch.cuda.synchronize()
            with ops.graph_capture(g, pool=self.pool):
                for c in range(cur, cur + n):
                    self._run(c, strip, mode, False)
            self.cache.len = keep                      # capture does not execute: replay below performs the steps
            hit = (g, self.last_tt, self.last_pos)     # the last step's token-type / position outputs live in the session's history buffers
            self.graphs[key] = hit
            captured = True
        else:
            captured = False
        g, self.last_tt, self.last_pos = hit
        g.replay()
        if mode[0] == "beam" and not captured:         # the capture pass already walked the ping-pong of the reordered caches
            self._swap_caches(n)
        self.cache.len = cur + n - 1 - strip

    def _run(self, cur, strip, mode, prefill):
        m = self.model
        kind, special, mask_token_id, top_k, temperature, eos, pad, train, top_p = mode
        if prefill:
            fed = self.ids[:, strip:cur]
            new, mask, tt, pos = m._step_inputs(fed, special, mask_token_id, prefill=True)
            new, tt, pos = new.contiguous(), tt.contiguous(), None if pos is None else pos.contiguous()
        else:                                                      # one launch: last token, token type, position, mask (+ their history)
            pair = isinstance(special, tuple) and special and isinstance(special[0], (tuple, list))
            sp0, sp1 = (special[0], special[1]) if pair else (special, special)
            longi = m.kind == "longitudinal"
            step_args = (self.ids, strip, cur, list(sp0), list(sp1), [0, 1, 0, 1] if longi else None, self.B // 2 if pair else self.B,
                         mask_token_id if longi else -1, self.new_id, self.tt1, self.pos1 if longi else None,
                         self.mask8 if longi else None, self.tt_hist, self.pos_hist if longi else None)
            mask = self.mask8[:, :cur - strip] if longi else None
            if (m._dec.fused_step_ok(self.cache, self.B, self.enc16) and len(sp0) <= 4 and len(sp1) <= 4
                    and cur - strip <= min(512, m.config.decoder.max_position_embeddings)):      # (the kernel indexes the position table with cur - strip - 1)
                # ... and the embeddings of the new token in the same launch, written in the decode kernels' activation layout
                ph, _, _ = m._dec._dropout_cfg(train, self.seed)
                x0 = ops.decode_step_embed(*step_args, *m._dec.embed_tables(), drop=(ph, self.seed, 1) if ph else None)
                logits = m._dec.decode_embedded(self.cache, x0, self.B, self.enc_mask8, mask, train=train, seed=self.seed)
                new = None
            else:
                ops.decode_step_inputs(*step_args)
                new, tt = self.new_id, self.tt1
                pos = self.pos1 if longi else None
        if new is not None:
            logits = m._dec.decode(self.cache, new, self.enc16, self.enc_mask8, mask, tt, pos, train=train, seed=self.seed)
        if kind == "beam":                                          # one launch: scoring, candidate search, bookkeeping; one more: cache reorder
            nb, penalty, (prompt_len, max_length) = top_k, temperature, top_p
            st, c = self.beam, self.cache
            ops.beam_step(logits, self.ids.view(nb, self.B // nb, self.Lmax), st["sequences"], st["run_scores"], st["beam_scores"], st["finished"],
                          st["unsat"], st["allhit"], st["beam_idx"], cur, max_length, -1 if eos is None else eos,
                          float(cur + 1 - prompt_len) ** penalty, ws=st["ws"])
            ops.gather_batch_multi(c.k + c.v, st["beam_idx"], c.len, c.k2 + c.v2)
            self._swap_caches()
            self.last_tt = self.last_pos = None
            return
        unf = self.unfinished if eos is not None else None
        eos_ = eos if eos is not None else -1
        n_smp = {"greedy": 0, "sample": self.B, "pair": self.B // 2}[kind]      # rows [0, n_smp) sample, the rest take the argmax
        col = self.ids[:, cur]                                                  # the selection kernels write straight into the id buffer
        if n_smp:                                                               # one launch for sampled + greedy rows
            u = self.u_all[cur, :n_smp]
            ops.select_token(logits, mode=1, temperature=temperature, top_k=top_k or 0, u=u, unfinished=unf, eos=eos_, pad=pad or 0, out=col,
                             top_p=top_p, n_sample=n_smp)
        else:
            ops.select_token(logits, unfinished=unf, eos=eos_, pad=pad or 0, out=col)
        if prefill:
            self.last_tt, self.last_pos = tt, pos
        else:                                                      # views of the history columns this step wrote (stable across graph replays)
            self.last_tt = self.tt_hist[:, cur:cur + 1]
            self.last_pos = self.pos_hist[:, cur:cur + 1] if m.kind == "longitudinal" else None


class GenerationMixin:
    graph_decode = True      # replay single-token decode steps from hipGraphs (DecodeSession)
    device_beam_search = True   # beam search with device-side bookkeeping (False: the host-loop restatement `_beam_search`, kept as an A/B check)

    def _session(self, B, S, Lmax, has_mask, share=1):
        cache = self.__dict__.setdefault("_decode_sessions", {})
        key = (B, S, Lmax, has_mask, share)
        if key not in cache:
            if len(cache) >= 4:
                cache.clear()
            cache[key] = DecodeSession(self, B, S, Lmax, has_mask, share)
        return cache[key]

    # ------------------------------------------------------------------------------------------ per-step inputs
    def _fed(self, ids, bos_token_id):
        if self.kind == "longitudinal" and bool(torch.all(ids[:, 0] == bos_token_id)):
            return ids[:, 1:]
        return ids

    def _step_inputs(self, fed, special_token_ids, mask_token_id, prefill):
        """-> (ids_new, attention_mask_full | None, token_type_ids, position_ids | None). `special_token_ids` is a list of ids, or a
        tuple of two such lists: the separator sets of the first and second half of the rows (sample + greedy decoded as one batch)."""
        sections = [0, 1, 0, 1] if self.kind == "longitudinal" else None
        fedc = fed.contiguous()
        if isinstance(special_token_ids, tuple) and special_token_ids and isinstance(special_token_ids[0], (tuple, list)):
            h = fedc.shape[0] // 2
            tt = torch.cat([ops.token_type_ids(fedc[:h], list(special_token_ids[0]), sections, past=not prefill),
                            ops.token_type_ids(fedc[h:], list(special_token_ids[1]), sections, past=not prefill)], dim=0)
        else:
            tt = ops.token_type_ids(fedc, list(special_token_ids), sections, past=not prefill)
        if self.kind == "longitudinal":
            mask, pos = ops.mask_position_ids(fedc, mask_token_id)
            return (fed, mask, tt, pos) if prefill else (fed[:, -1:], mask, tt, pos[:, -1:])
        return (fed, None, tt, None) if prefill else (fed[:, -1:], None, tt, None)

    # ------------------------------------------------------------------------------------------ generate
    def _generate(self, inputs=None, pixel_values=None, encoder_outputs=None, decoder_input_ids=None, input_ids=None,
                  special_token_ids=None, mask_token_id=None, max_length=None, max_new_tokens=None, bos_token_id=None,
                  eos_token_id=None, pad_token_id=None, num_beams=1, do_sample=False, top_k=50, top_p=1.0, temperature=1.0,
                  output_scores=False, return_dict_in_generate=False, use_cache=True, length_penalty=1.0, return_margins=False,
                  forced_tokens=None, record_inputs=False, **unused):
        from .modelling import ModelOutput
        if inputs is not None and pixel_values is None:
            pixel_values = inputs
        if special_token_ids is None:
            raise ValueError("special_token_ids is required (reference prepare_inputs_for_generation signature)")
        if self.kind == "longitudinal" and mask_token_id is None:
            raise ValueError("mask_token_id is required for the longitudinal model")
        top_p = 1.0 if top_p is None else float(top_p)
        with torch.no_grad():
            if encoder_outputs is None:
                encoder_outputs = self._encode(pixel_values)
        enc = encoder_outputs[0]
        enc_mask = encoder_outputs.get("attention_mask") if isinstance(encoder_outputs, dict) else None
        if self.kind == "single":
            enc_mask = None
        dev = self.device
        B = enc.shape[0]
        prompt = decoder_input_ids if decoder_input_ids is not None else input_ids
        start = torch.full((B, 1), bos_token_id, dtype=torch.int64, device=dev)
        if prompt is None:
            ids = start
        else:
            prompt = prompt.to(device=dev, dtype=torch.int64)
            ids = torch.cat([start, prompt], dim=-1) if bool((prompt[:, 0] != bos_token_id).all()) else prompt
        if max_new_tokens is not None:
            max_length = ids.shape[1] + max_new_tokens
        if max_length is None:
            max_length = 20 + ids.shape[1]
        enc16 = (enc if enc.dtype == torch.bfloat16 else ops.cast_to_bf16(enc.float().contiguous())).detach().contiguous()
        enc_mask8 = None if enc_mask is None else enc_mask.to(device=dev, dtype=torch.uint8).contiguous()
        if num_beams > 1:
            if do_sample:
                raise NotImplementedError("beam sampling is not used by the reference")
            with torch.no_grad():
                # device-side search: beam counts with a shared-K/V attention kernel, and a cache the one-launch reorder can address (16 tensors = 8 layers)
                on_device = self.device_beam_search and num_beams in (2, 4) and 2 * self.config.decoder.num_hidden_layers <= 16
                search = self._beam_search_session if on_device else self._beam_search
                seqs, seq_scores = search(ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, num_beams,
                                          bos_token_id, eos_token_id, pad_token_id, length_penalty)
            if return_dict_in_generate:
                return ModelOutput(sequences=seqs, sequences_scores=seq_scores if output_scores else None)
            return seqs

        prompt_len = ids.shape[1]
        rec = {"tt": [], "pos": []}
        margins = []
        plain = forced_tokens is None and not return_margins and not record_inputs
        # ---- the SCST caller's pair of decodes (reference scst/gt_prompt.py:162-180 then :94-112): a sampling call with scores, followed by a
        # greedy call over the SAME encoder outputs and prompt. A cached decode step costs the same for 16 and for 32 rows (it streams the
        # decoder weights), so the sampling call decodes the greedy rows too -- with the arguments the greedy call of the PREVIOUS step used --
        # and the greedy call that follows, if it asks for exactly that, gets the finished rows (`speculative_baseline = False` switches it off).
        # the speculative result is tied to the very tensor OBJECTS of the sampling call (held alive here, so the allocator cannot hand their
        # addresses to another batch), their in-place version counters, and the weights' state: FusedAdamW moves shadow_version, torch.optim's
        # in-place updates move the version counter the parameter views share with the flat master buffer
        spec_objs = (enc, prompt, enc_mask)
        spec_ver = (enc._version, None if prompt is None else prompt._version, None if enc_mask is None else enc_mask._version,
                    getattr(self, "shadow_version", 0), self.flat32._version, bool(self.training))
        if plain and not do_sample and prompt is not None:
            want = (tuple(special_token_ids), mask_token_id, int(max_length), bos_token_id, eos_token_id, pad_token_id)
            hit = getattr(self, "_spec_result", None)
            self._spec_result = None
            self._spec_pattern = want                            # what the next sampling call may decode along
            if (hit is not None and all(a is b for a, b in zip(hit[0], spec_objs)) and hit[3] == spec_ver and hit[1] == want
                    and getattr(self, "speculative_baseline", True)):
                return ModelOutput(sequences=hit[2], scores=None) if return_dict_in_generate else hit[2]
        pattern = getattr(self, "_spec_pattern", None)
        if (plain and do_sample and output_scores and prompt is not None and pattern is not None and getattr(self, "speculative_baseline", True)
                and pattern[2] == int(max_length) and pattern[1] == mask_token_id and pattern[3:] == (bos_token_id, eos_token_id, pad_token_id)):
            with torch.no_grad():
                ids, base, rec = self.sample_and_greedy(encoder_outputs, prompt, list(special_token_ids), list(pattern[0]), mask_token_id, int(max_length),
                                                        bos_token_id, eos_token_id, pad_token_id, top_k=top_k, temperature=temperature, top_p=top_p)
            self._spec_result = (spec_objs, pattern, base, spec_ver)
        elif forced_tokens is None and not return_margins:
            ids = self._generate_session(ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, bos_token_id, eos_token_id,
                                         pad_token_id, do_sample, top_k, temperature, rec if ((output_scores and do_sample) or record_inputs) else None,
                                         top_p=top_p)
        else:
            ids, margins = self._generate_eager(ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, bos_token_id, eos_token_id,
                                                pad_token_id, forced_tokens, return_margins,
                                                rec if ((output_scores and do_sample) or record_inputs) else None)

        scores = None
        if output_scores and do_sample:
            scores = self._rescore_sampled(ids, prompt_len, rec, enc, enc_mask, special_token_ids, mask_token_id, bos_token_id,
                                           top_k, temperature, top_p)
        if return_dict_in_generate:
            out = ModelOutput(sequences=ids, scores=scores)
            if record_inputs:
                out["recorded_inputs"] = rec
            if margins:
                out["greedy_tokens"] = torch.stack([m[0] for m in margins], 1)
                out["greedy_margins"] = torch.stack([m[1] for m in margins], 1)
            return out
        return ids

    def _generate_eager(self, ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, bos_token_id, eos_token_id, pad_token_id,
                        forced_tokens, return_margins, rec=None):
        """Eagerly launched greedy loop with per-step argmax / margin capture and optional teacher forcing (parity tests). rec: the per-step
        token-type / position inputs are recorded as _generate_session records them (a sampling call with forced_tokens then returns the processed
        scores of the FORCED ids: the reference's sampled sequence can be pushed through the grad-enabled generate body)."""
        dev = self.device
        B = ids.shape[0]
        prompt_len = ids.shape[1]
        margins = []
        with torch.no_grad():
            cache = self._dec.new_cache(B, max_length, dev)
            unfinished = torch.ones(B, dtype=torch.int32, device=dev)
            step = 0
            seed = self.next_dropout_seed() if self.training else None
            while ids.shape[1] < max_length:
                fed = self._fed(ids, bos_token_id)
                new, mask, tt, pos = self._step_inputs(fed, special_token_ids, mask_token_id, prefill=cache.len == 0)
                if rec is not None:
                    rec["tt"].append(tt.clone())
                    rec["pos"].append(None if pos is None else pos.clone())
                    rec["seed"] = seed
                logits = self._dec.decode(cache, new.contiguous(), enc16, enc_mask8, mask, tt.contiguous(), None if pos is None else pos.contiguous(),
                                          seed=seed)
                if forced_tokens is not None:
                    greedy_tok, mg = ops.select_token(logits, need_margin=True)
                    margins.append((greedy_tok, mg))
                    nxt = forced_tokens[:, step].to(dev).clone()
                    if eos_token_id is not None:
                        nxt = torch.where(unfinished.bool(), nxt, torch.full_like(nxt, pad_token_id))
                        unfinished = unfinished & (nxt != eos_token_id).int()
                else:
                    nxt, mg = ops.select_token(logits, unfinished=unfinished if eos_token_id is not None else None,
                                               eos=eos_token_id if eos_token_id is not None else -1, pad=pad_token_id or 0, need_margin=True)
                    margins.append((nxt.clone(), mg))
                ids = torch.cat([ids, nxt[:, None]], dim=-1)
                step += 1
                if forced_tokens is not None and step >= forced_tokens.shape[1]:
                    break
                if eos_token_id is not None and forced_tokens is None and int(unfinished.max()) == 0:
                    break
        return ids, margins

    def _generate_session(self, ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, bos_token_id, eos_token_id, pad_token_id,
                          do_sample, top_k, temperature, rec, top_p=1.0):
        """Greedy / top-k sampling over a DecodeSession (static buffers, graph-replayed steps, EOS polled every 8 steps)."""
        dev = self.device
        B, prompt_len = ids.shape
        strip = 1 if (self.kind == "longitudinal" and bool(torch.all(ids[:, 0] == bos_token_id))) else 0
        with torch.no_grad():
            ses = self._session(B, enc16.shape[1], max_length, enc_mask8 is not None, share=B // enc16.shape[0])
            ses.reset(ids, enc16, enc_mask8)
            if do_sample == "pair":
                kind, special = "pair", (tuple(special_token_ids[0]), tuple(special_token_ids[1]))
            else:
                kind, special = ("sample" if do_sample else "greedy"), tuple(special_token_ids)
            mode = (kind, special, mask_token_id, int(top_k or 0), float(temperature), eos_token_id, pad_token_id, bool(self.training), float(top_p))
            if rec is not None:
                rec["seed"] = ses.seed.clone() if self.training else None
                # the session's cross-attention K / V of every layer (valid until the session's next decode) and the weight version they belong to
                rec["cross_kv"] = None if ses.cache.kv_all is None else (ses.cache.kv_all, tuple(ses.enc16.shape), self._weights_stamp(), ses, ses.fills)
            cur = prompt_len
            first_tt = first_pos = None
            poll = None                      # (pinned host word, event): "any row unfinished?" as of the PREVIOUS poll -- read without stalling the queue
            while cur < max_length:
                # the prefill step alone, then chunks of up to 8 cached steps per graph replay, aligned to the EOS polling period
                n = 1 if cur == prompt_len else min(8 - (cur - prompt_len) % 8, max_length - cur)
                ses.step(cur, strip, mode, n)
                if rec is not None and cur == prompt_len:            # the prefill step's inputs cover the whole prompt
                    first_tt, first_pos = ses.last_tt.clone(), None if ses.last_pos is None else ses.last_pos.clone()
                cur += n
                if eos_token_id is not None and ((cur - prompt_len) % 8 == 0):
                    # EOS polling every 8 steps, one poll behind: the copy of this poll's flag is only waited for at the NEXT poll, so the host
                    # keeps enqueueing steps while the GPU works (a blocking .item() drained the queue 32 times per decode); the up to 16 steps of
                    # overshoot are trimmed below, as the 8 of a blocking poll were
                    if poll is not None:
                        poll[1].synchronize()
                        if int(poll[0][0]) == 0:
                            break
                    host = ses.poll_host
                    host.copy_(ses.unfinished.max().reshape(1), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    poll = (host, ev)
            out = ses.ids[:, :cur].clone()
            if rec is not None:
                # cached steps appended their token type / position to the session's history columns: ONE copy, then per-step views
                # (the session is reused by the next generate() call, so the history must not be handed out itself)
                rec["tt"].append(first_tt)
                rec["pos"].append(first_pos)
                if cur > prompt_len + 1:
                    th = ses.tt_hist[:, prompt_len + 1:cur].clone()
                    ph_ = ses.pos_hist[:, prompt_len + 1:cur].clone() if first_pos is not None else None
                    for c in range(th.shape[1]):
                        rec["tt"].append(th[:, c:c + 1])
                        rec["pos"].append(None if ph_ is None else ph_[:, c:c + 1])
            # HF stops at the step on which the last row finishes: trim what the 8-step polling overshot
            if eos_token_id is not None:
                gen = out[:, prompt_len:]
                is_eos = gen == eos_token_id
                first = torch.where(is_eos.any(1), is_eos.int().argmax(1) + 1, torch.full((B,), gen.shape[1], device=dev))
                out = out[:, : prompt_len + int(first.max())]
        return out

    @torch.no_grad()
    def _weights_stamp(self):
        """Changes whenever the decoder weights may have: the bf16 shadow's version (fused AdamW) and the fp32 master's autograd version (torch edits)."""
        return (getattr(self, "shadow_version", 0), self.flat32._version)

    def _session_cross_kv(self, rec, enc, own=False):
        """The cross-attention K / V of all layers a decode session projected at prefill (rec["cross_kv"]), if they still belong to `enc`'s shape and
        to the current weights -- the teacher-forced re-scoring pass of the same SCST step then skips its own projection GEMM. Else None.
        The result is a LOAN of the session's static buffer (decoder.SharedCrossKV): valid until that session's next prefill. own=True hands out a
        private copy instead (one 0.15-ms copy at the benchmark shape instead of a 16-GFLOP projection) -- for the autograd bridge, whose backward runs
        whenever the caller gets to it: the reference's SCST caller decodes its greedy baseline on the same geometry BETWEEN the sampling call and
        loss.backward() (scst/gt_prompt.py:84-132)."""
        ckv = rec.get("cross_kv") if rec else None
        if ckv is None or ckv[1] != tuple(enc.shape) or ckv[2] != self._weights_stamp():
            return None
        from .decoder import SharedCrossKV
        loan = SharedCrossKV(ckv[0], ckv[3], ckv[4])
        if not loan.valid():
            return None                                    # the session has decoded another batch since: project again
        return SharedCrossKV(ckv[0].clone(), None, 0) if own else loan

    def sample_and_greedy(self, encoder_outputs, prompt_ids, special_sample, special_greedy, mask_token_id, max_length, bos_token_id,
                          eos_token_id, pad_token_id, top_k=50, temperature=1.0, top_p=1.0):
        """The two decodes of one SCST step (reference scst/gt_prompt.py:162-180 sample, :94-112 greedy baseline) as ONE batch of 2B
        rows over the same studies: a cached decode step is bound by streaming the decoder weights, so both halves share every weight
        read and every launch. Rows [0,B) are sampled (top-k, temperature) with the separator set `special_sample`, rows [B,2B) take the
        argmax with `special_greedy`; each half sees exactly the per-step inputs its own generate() call would have produced.
        -> (sampled sequences [B,L], greedy sequences [B,L], recorded per-step token-type / position ids of the sampled half)."""
        enc = encoder_outputs[0]
        enc_mask = encoder_outputs.get("attention_mask") if isinstance(encoder_outputs, dict) else None
        if self.kind == "single":
            enc_mask = None
        dev = self.device
        B = enc.shape[0]
        prompt = prompt_ids.to(device=dev, dtype=torch.int64)
        start = torch.full((B, 1), bos_token_id, dtype=torch.int64, device=dev)
        first = prompt[:, 0].tolist()                                        # (the one host read of this call: the prompt's first column)
        prepend = all(v != bos_token_id for v in first)
        ids = torch.cat([start, prompt], dim=-1) if prepend else prompt
        enc16 = (enc if enc.dtype == torch.bfloat16 else ops.cast_to_bf16(enc.float().contiguous())).detach().contiguous()
        enc_mask8 = None if enc_mask is None else enc_mask.to(device=dev, dtype=torch.uint8).contiguous()      # B rows serve the 2B decode rows
        rec = {"tt": [], "pos": []}
        out = self._generate_session(torch.cat([ids, ids], dim=0), enc16, enc_mask8, (special_sample, special_greedy), mask_token_id,
                                     max_length, bos_token_id, eos_token_id, pad_token_id, "pair", top_k, temperature, rec, top_p=top_p)
        rec = {"tt": [t[:B] for t in rec["tt"]], "pos": [None if p_ is None else p_[:B] for p_ in rec["pos"]], "seed": rec.get("seed"),
               "cross_kv": rec.get("cross_kv")}

        def trim(seq):                                   # each half ends where ITS last row finished (HF stops per generate() call)
            if eos_token_id is None:
                return seq
            gen = seq[:, ids.shape[1]:]
            is_eos = gen == eos_token_id
            first = torch.where(is_eos.any(1), is_eos.int().argmax(1) + 1, torch.full((B,), gen.shape[1], device=dev))
            return seq[:, : ids.shape[1] + int(first.max())]
        return trim(out[:B]), trim(out[B:]), rec

    generate = torch.no_grad()(_generate)          # `.generate.__wrapped__` is the grad-enabled body (reference scst/gt_prompt.py:162)

    # ------------------------------------------------------------------------------------------ differentiable scores
    def _rescore_sampled(self, ids, prompt_len, rec, enc, enc_mask, special_token_ids, mask_token_id, bos_token_id, top_k, temperature, top_p=1.0):
        """Processed scores of every sampling step, [B,V] each, with autograd through the decoder (and encoder_outputs)."""
        fed = self._fed(ids, bos_token_id)
        stripped = ids.shape[1] - fed.shape[1]
        n_new = ids.shape[1] - prompt_len
        steps = len(rec["tt"])
        n_new = min(n_new, steps)
        tf_in = fed[:, : (prompt_len - stripped) + n_new - 1]
        tt = torch.cat(rec["tt"][:n_new], dim=1)
        pos = torch.cat(rec["pos"][:n_new], dim=1) if rec["pos"][0] is not None else None
        mask = (tf_in != mask_token_id).to(torch.uint8) if self.kind == "longitudinal" else None
        # the LM head runs on the sampled positions only: sc [B, n_new, V] is the (contiguous) output itself, and its gradient goes back as it is --
        # a slice of full-length logits costs a 490-MB zero fill + copy in backward at the benchmark shape
        first = prompt_len - stripped - 1
        sc = self._decode_tf(tf_in, enc, None if self.kind == "single" else enc_mask, mask, tt, pos, seed=rec.get("seed"),
                             cross_kv=self._session_cross_kv(rec, enc, own=True), logit_from=first)
        if temperature is not None and float(temperature) != 1.0:
            sc = sc / float(temperature)
        if top_k or top_p < 1.0:
            with torch.no_grad():                                            # sc is already temperature-scaled: warpers see temperature 1
                flat = sc.detach().reshape(-1, sc.shape[-1]).contiguous()
                thr = ops.topk_threshold(flat, int(top_k or 0), top_p, 1.0).view(sc.shape[0], sc.shape[1], 1)
            # The token that WAS sampled at a step is inside that step's kept set by construction (reference: the scores are the ones it was
            # drawn from). These scores are recomputed by a teacher-forced pass whose logits differ from the cached step's in the last bf16
            # bits, so a token drawn at the very edge of the top-k can fall just below the recomputed threshold (a few per 4080 draws at the
            # benchmark shape). It keeps its score -- a -inf there would turn the caller's REINFORCE loss into inf - inf -- and takes the place of
            # the k-th entry, so the row still has top_k finite entries, as the reference's rows do (same rule as csrc/loss.hip kept_threshold).
            drawn = ids[:, prompt_len: prompt_len + n_new].unsqueeze(-1).clamp(min=0)
            with torch.no_grad():
                below = sc.detach().gather(2, drawn) < thr
                thr = torch.where(below, torch.nextafter(thr, torch.full_like(thr, float("inf"))), thr)
            drop = sc < thr                                                  # TopK (+ TopP) LogitsWarper semantics (ties at the threshold kept)
            drop = drop.scatter(2, drawn, False)
            sc = sc.masked_fill(drop, float("-inf"))
        # one unbind (its backward is ONE stack of the steps' gradients; 255 separate slices each scattered their gradient into a zero tensor of
        # the full [B, T, V] size: 85 ms per SCST step at the benchmark shape). Each step remembers where it came from so that the caller's
        # `torch.stack(scores, dim=-1)` is a view of `sc` again (modelling.BoundaryTensor).
        from .modelling import _as_boundary
        sc = sc.contiguous()
        return tuple(_as_boundary(s_t, kind="step", base=sc, t=t) for t, s_t in enumerate(sc.unbind(1)))

    # ------------------------------------------------------------------------------------------ beam search
    def _beam_search_session(self, ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, num_beams, bos, eos, pad, length_penalty):
        """Beam search (TF5 generation/utils.py:3208-3560, do_sample=False, early_stopping=False, num_return_sequences=1) over a DecodeSession:
        beams*B rows decoded by the cached-step kernels (the beams of a study share its cross-attention K/V: ONE copy of the encoder rows, as in
        the SCST sample+greedy batch), the whole per-step bookkeeping in one kernel launch (ops.beam_step), the cache reorder in another, the
        steps replayed from hipGraphs eight at a time, and the stop condition polled asynchronously every 8 steps (the stopped state is frozen
        on the device, so overshoot steps change nothing). The host loop `_beam_search` does the same with ~40 tensor ops and two
        synchronisations per token."""
        B, prompt_len = ids.shape
        nb = num_beams
        pad = 0 if pad is None else pad
        strip = 1 if (self.kind == "longitudinal" and bool(torch.all(ids[:, 0] == bos))) else 0
        with torch.no_grad():
            ses = self._session(B * nb, enc16.shape[1], max_length, enc_mask8 is not None, share=nb)
            ses.reset(ids.repeat(nb, 1), enc16, enc_mask8)
            ses.ids[:, prompt_len:] = pad
            st = ses.beam_state(nb, pad)
            mode = ("beam", tuple(special_token_ids), mask_token_id, nb, float(length_penalty), eos, pad, bool(self.training), (prompt_len, max_length))
            cur = prompt_len
            poll = None
            while cur < max_length:
                n = 1 if cur == prompt_len else min(8 - (cur - prompt_len) % 8, max_length - cur)
                ses.step(cur, strip, mode, n)
                cur += n
                if (cur - prompt_len) % 8 == 0 and cur < max_length:
                    if poll is not None:
                        poll[1].synchronize()
                        if int(poll[0][0]) == 0:
                            break
                    par = (cur - 1) & 1                                     # flags of the last executed step (column cur - 1)
                    going = ((st["unsat"][par].max() > 0) & (st["allhit"][par].min() == 0)).to(torch.int32).reshape(1)
                    ses.poll_host.copy_(going, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    poll = (ses.poll_host, ev)
            out = st["sequences"][0].clone()
            scores = st["beam_scores"][:, 0].clone()
        if eos is None:
            return out, scores
        gen = out[:, prompt_len:]
        is_eos = gen == eos
        lens = torch.where(is_eos.any(1), is_eos.int().argmax(1) + 1, torch.full((B,), gen.shape[1], device=out.device))
        return out[:, : prompt_len + int(lens.max())], scores

    def _beam_search(self, ids, enc16, enc_mask8, special_token_ids, mask_token_id, max_length, num_beams, bos, eos, pad, length_penalty):
        """TF5 generation/utils.py:3208-3560 with do_sample=False, early_stopping=False, one EOS id, num_return_sequences=1."""
        dev = ids.device
        B = ids.shape[0]
        nb, keep = num_beams, 2 * num_beams
        V = self.config.decoder.vocab_size
        ids = ids.repeat_interleave(nb, dim=0)
        enc16 = enc16.repeat_interleave(nb, dim=0).contiguous()
        enc_mask8 = None if enc_mask8 is None else enc_mask8.repeat_interleave(nb, dim=0).contiguous()
        cur = prompt_len = ids.shape[1]
        running = torch.full((B, nb, max_length), pad, dtype=torch.int64, device=dev)
        running[:, :, :cur] = ids.view(B, nb, cur)
        sequences = running.clone()
        run_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
        run_scores[:, 1:] = -1e9
        beam_scores = torch.full((B, nb), -1e9, dtype=torch.float32, device=dev)
        finished = torch.zeros((B, nb), dtype=torch.bool, device=dev)
        unsat = torch.ones((B, 1), dtype=torch.bool, device=dev)
        top_mask = torch.arange(keep, device=dev) < nb
        batch_off = (torch.arange(B, device=dev) * nb).view(B, 1)
        cache = self._dec.new_cache(B * nb, max_length, dev)
        seed = self.next_dropout_seed() if self.training else None

        def gather(t, idx):
            while idx.dim() < t.dim():
                idx = idx.unsqueeze(-1)
            return torch.gather(t, 1, idx.expand(-1, -1, *t.shape[2:]))

        while True:
            flat = running[:, :, :cur].reshape(B * nb, cur)
            fed = self._fed(flat, bos)
            new, mask, tt, pos = self._step_inputs(fed, special_token_ids, mask_token_id, prefill=cache.len == 0)
            logits = self._dec.decode(cache, new.contiguous(), enc16, enc_mask8, mask, tt.contiguous(), None if pos is None else pos.contiguous(),
                                      seed=seed)
            ops.log_softmax_rows_(logits, add_row=run_scores.view(-1).contiguous())          # log_softmax + running beam score
            topk_lp, topk_idx = ops.topk_rows(logits.view(B, nb * V), keep)
            beam_of = topk_idx // V
            topk_seq = gather(running, beam_of)
            topk_seq[:, :, cur] = topk_idx % V
            hits = (topk_seq[:, :, cur] == (-1 if eos is None else eos)) | (cur + 1 >= max_length)
            run_lp = topk_lp + hits.float() * -1.0e9
            nxt = torch.topk(run_lp, k=nb)[1]
            running, run_scores = gather(topk_seq, nxt), gather(run_lp, nxt)
            beam_idx = (gather(beam_of, nxt) + batch_off).view(-1)
            just = hits & top_mask[None, :]
            fin_lp = topk_lp / ((cur + 1 - prompt_len) ** length_penalty)
            fin_lp = fin_lp + (~unsat).float() * -1.0e9 + (~just).float() * -1.0e9
            m_seq = torch.cat((sequences, topk_seq), dim=1)
            m_sc = torch.cat((beam_scores, fin_lp), dim=1)
            m_fin = torch.cat((finished, just), dim=1)
            best = torch.topk(m_sc, k=nb)[1]
            sequences, beam_scores, finished = gather(m_seq, best), gather(m_sc, best), gather(m_fin, best)
            cache.reorder(beam_idx.contiguous())
            cur += 1
            best_run = run_scores[:, :1] / ((cur - prompt_len) ** length_penalty)
            worst_fin = torch.where(finished, beam_scores.min(dim=1, keepdim=True)[0], torch.full_like(beam_scores, -1.0e9))
            unsat = unsat & torch.any(best_run > worst_fin, dim=-1, keepdim=True)
            if not (bool(unsat.any()) and not bool(hits.all())):
                break
        out = sequences[:, 0, :]
        if eos is None:
            return out, beam_scores[:, 0]
        gen = out[:, prompt_len:]
        is_eos = gen == eos
        lens = torch.where(is_eos.any(1), is_eos.int().argmax(1) + 1, torch.full((B,), gen.shape[1], device=dev))
        return out[:, : prompt_len + int(lens.max())], beam_scores[:, 0]
