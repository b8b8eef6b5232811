"""Self-critical sequence training step on MI355X -- the engine-side analogue of SCSTGTPrompt.scst_step
(reference modules/lightning_modules/longitudinal/scst/gt_prompt.py:62-142, :144-209 sample, :211-246 reinforce_loss).

    1  encoder (frozen, no grad)                                     :84
    2  sample with top-k=50 / T=1 multinomial, KV-cached             :162-180   special_token_ids = [bos, sep]           (quirk Q5)
    3  reward(sample)                                                :90-91
    4  greedy baseline, KV-cached                                    :94-112    special_token_ids = [pmt-sep, bos, sep]
    5  reward(baseline); advantage = reward - baseline               :126-129
    6  REINFORCE: nll of the sampled ids under the top-k-filtered distribution, summed over time, weighted by the advantage,
       averaged over the batch                                       :230-244
    7  backward through the decoder, gradient all-reduce, AdamW

Instead of keeping an autograd graph alive across 255 cached decode steps, step 6 re-scores the sampled sequence with ONE
teacher-forced pass whose per-position inputs (token types, positions, masks) are exactly the ones the cached steps used, and the
fused loss kernel applies the top-k threshold, log-softmax, nll, advantage weighting and emits d(logits) in one pass.
"""
from __future__ import annotations

import torch

from . import dp, ops


_SIDE = {}
TRACE = None            # set to a list (scripts/r5/scst_timeline.py): scst_step appends (label, host seconds, CUDA event) at its phase boundaries


def _mark(label):
    if TRACE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        TRACE.append((label, __import__("time").perf_counter(), ev))

def _host_copies(tensors):
    """Asynchronous device -> pinned-host copies on a side stream that waits only for what is queued so far (the decode): the main stream
    goes on with the re-scoring forward while the copies land. -> (list of pinned CPU tensors, event to synchronise before reading them)."""
    dev = tensors[0].device
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    outs = []
    with torch.cuda.stream(side):
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            t.record_stream(side)
            outs.append(h)
        done = torch.cuda.Event()
        done.record(side)
    return outs, done


def scst_step(model, opt, reward_fn, images, prompt_ids, label_texts, special, decoder_max_len, top_k=50, temperature=1.0,
              fused_decode=True, top_p=1.0, reward_on_host=False):
    """special = dict(bos, eos, sep, pad, pmt_sep). reward_fn -> fp32 [B] given the caller's labels (closure):
      * reward_on_host=False: reward_fn(sequences_without_prompt [B,L] int64 on the device) -- benchmarks feed synthetic re-tokenised ids;
      * reward_on_host=True: reward_fn(full sequences [B, P+L] int64 in PINNED HOST memory) -- the reference's real path (gt_prompt.py:192-197,
        120-128: split_and_decode_sections -> strings -> CXR-BERT tokenizer), e.g. reward.ReportReward. The device -> host copies run on a side
        stream and the teacher-forced re-scoring forward is queued BEFORE the rewards are needed, so the GPU works through it while the CPU
        decodes and re-tokenises the reports.
    Returns dict(loss, reward, baseline, seq_len)."""
    bos, eos, sep, pad, pmt_sep = (special[k] for k in ("bos", "eos", "sep", "pad", "pmt_sep"))
    temperature = 1.0 if temperature is None else float(temperature)
    if temperature <= 0.0:
        raise ValueError("temperature must be positive")
    dev = model.device
    P = prompt_ids.shape[1]
    with torch.no_grad():
        _mark("start")
        eo = model.encoder(images)
        _mark("encoder queued")
        if fused_decode:
            # sample + greedy baseline as one 2B-row cached decode (same per-row inputs as the two separate calls below)
            seqs, base, rec = model.sample_and_greedy(eo, prompt_ids, [bos, sep], [pmt_sep, bos, sep], pad, decoder_max_len + P, bos, eos, pad,
                                                      top_k=top_k, temperature=temperature, top_p=top_p)
        else:
            smp = model.generate(input_ids=prompt_ids, special_token_ids=[bos, sep], encoder_outputs=eo, bos_token_id=bos, eos_token_id=eos,
                                 pad_token_id=pad, mask_token_id=pad, return_dict_in_generate=True, do_sample=True, num_beams=1, use_cache=True,
                                 top_k=top_k, top_p=top_p, temperature=temperature, max_new_tokens=decoder_max_len - 1, record_inputs=True)
            seqs, rec = smp["sequences"], smp["recorded_inputs"]
            base = model.generate(encoder_outputs=eo, decoder_input_ids=prompt_ids, special_token_ids=[pmt_sep, bos, sep],
                                  max_length=decoder_max_len + P, bos_token_id=bos, eos_token_id=eos, pad_token_id=pad, mask_token_id=pad,
                                  num_beams=1, return_dict_in_generate=True, use_cache=True)["sequences"]
        # strip the leading BOS column (gt_prompt.py:185-186): read back from the sequences (one host round trip at the end of the decode). Queueing the
        # rest of the step WITHOUT that read (the fused decode knows from the prompt whether the column is there) was measured in round 5 and removed in
        # round 6: beside the weight-gradient stream the part of the step behind the decode then takes ~9.5 ms LONGER (103.9 -> 113.0 ms)
        if bool(torch.all(seqs[:, 0] == bos)):
            seqs = seqs[:, 1:]
        if bool(torch.all(base[:, 0] == bos)):
            base = base[:, 1:]
        sampled = seqs[:, P:].contiguous()
        _mark("decode done (host has read the sequences)")
        host = ticket = None
        if reward_on_host:
            host = _host_copies([seqs.contiguous(), base.contiguous()])
            start = getattr(reward_fn, "pair_start", None)
            if start is not None:
                # the decode is complete here (the BOS check above read the sequences back), so the copies land within ~0.1 ms: hand the ids to the
                # reward's string worker NOW -- a child process turns them into strings and reward-tokenizer ids while this thread queues the
                # re-scoring pass below (the tokenizers library keeps the GIL: a thread of this process could not run beside the launches)
                host[1].synchronize()
                ticket = start(host[0][0], host[0][1])

        # ---- REINFORCE through one teacher-forced pass: the forward needs no reward, so it is queued first
        opt.zero_grad()
        n_new = sampled.shape[1]
        tf_in = seqs[:, : P + n_new - 1].contiguous()
        tt = torch.cat(rec["tt"][:n_new], dim=1).contiguous()
        pos = torch.cat(rec["pos"][:n_new], dim=1).contiguous()
        mask = (tf_in != pad).to(torch.uint8)
        enc = eo.last_hidden_state
        enc_mask = eo.attention_mask.to(torch.uint8).contiguous()
        # train mode: the same dropout seed as the cached sampling decode -> the re-scored network IS the one that sampled
        # the decode session projected the cross-attention K / V of all layers from this encoder output at prefill: the re-scoring pass reads them
        # (the LM head runs on the n_new sampled positions only: logits [B, n_new, V], contiguous)
        logits, saved = model._dec.forward(tf_in, enc.contiguous(), enc_mask, mask, tt, pos, save=True, seed=rec.get("seed"),
                                           cross_kv=model._session_cross_kv(rec, enc), logit_from=P - 1)

        B, _, V = logits.shape                                               # scores of the n_new sampling steps
        flat = logits.view(-1, V)
        if temperature != 1.0:
            # scst_sample_temperature (gt_prompt.py:13-15,177): the processed scores are logits / T (TemperatureLogitsWarper runs first), so the loss is
            # taken on the scaled scores and d(logits) = d(scores) / T below; the warpers then see temperature 1, as in the drop-in generate body
            flat.mul_(1.0 / temperature)
        # (the warper threshold needs no reward either: queued in front of the host's string work)
        thr = ops.topk_threshold(flat, int(top_k or 0), top_p, 1.0) if (top_k or top_p < 1.0) else None
        _mark("re-scoring forward + threshold queued")
        # reward of the sampled and of the greedy reports. A reward_fn with `.pair(sampled, greedy)` scores both in ONE pass (one tokenizer call,
        # one 2B-row CXR-BERT forward: the two B-row forwards of a BERT-base are launch-bound, ~2.6 ms each)
        pair = getattr(reward_fn, "pair", None)
        if reward_on_host:
            (seqs_h, base_h), done = host
            done.synchronize()                                               # only the decode had to finish; the forward above keeps the GPU busy
            if ticket is not None:
                reward, baseline = reward_fn.pair_finish(ticket)
            else:
                reward, baseline = pair(seqs_h, base_h) if pair else (reward_fn(seqs_h), reward_fn(base_h))
        else:
            base_new = base[:, P:].contiguous()
            reward, baseline = pair(sampled, base_new) if pair else (reward_fn(sampled), reward_fn(base_new))
        _mark("rewards queued (strings done)")
        adv = (reward - baseline).float().contiguous()
        # data parallel: the sampled / greedy sequences and their rewards of ALL ranks (RCCL all-gather over xGMI, <= 64 KB per rank) for the
        # global reward / baseline statistics the step reports; the advantage above stays per study, as in the reference (gt_prompt.py:129-132)
        glob = dp.gather_scst_statistics(sampled, base[:, P:].contiguous(), reward, baseline, pad, max_sampled=decoder_max_len, max_greedy=decoder_max_len)
        labels = sampled.reshape(-1)
        w = ops.ce_weights(labels, pad, mode=1, reward=adv, T=n_new)
        loss, _, dl = ops.softmax_ce(flat, labels, pad, w, thr=thr)
        if temperature != 1.0:
            dl.mul_(1.0 / temperature)
        from .training import wgrad_overlap
        with wgrad_overlap():                                                # weight-gradient GEMMs beside the dX chain, as in the TF step
            model._dec.backward(saved, dlogits=dl, need_denc=False)
            ops.wgrad_join()
        _mark("backward queued + joined")
        world = dp.world_size()
        if dp.active():
            opt.reducer.reduce_range(0, model._param_total)
            opt.reducer.wait()
        opt.step(gscale=1.0 / world)
        _mark("optimiser queued")
        seq_len = (glob["sampled"] != pad).sum(-1).float().mean()
    # dropout_seed / encoder_seed: the device words the counter-based dropout / DropPath hashes of this step were keyed with (None in eval
    # mode) -- cxr_dropout_mask materialises the masks from them (parity tests hand them to the CPU oracle)
    return {"loss": loss, "reward": glob["reward"].mean(), "baseline": glob["baseline"].mean(), "seq_len": seq_len, "sampled": sampled,
            "baseline_ids": base, "global": glob, "dropout_seed": rec.get("seed"), "encoder_seed": getattr(model._enc, "_seed", None)}


def scst_generated_prompt_step(model, opt, reward, tokenizer, images, previous_findings, previous_impression, findings, impression,
                               decoder_max_len, top_k=50, top_p=1.0, temperature=1.0):
    """SCST step of the GENERATED-prompt model -- the engine-side analogue of SCSTGeneratedPrompt.scst_step (reference
    modules/lightning_modules/longitudinal/scst/gen_prompt.py:174-259, BASELINE.json configs[4]): the prompt is tokenised from the previous
    study's GENERATED findings / impression strings (`tokenize_prompt(..., add_bos_token_id=True)`, :186-192), the step is the gt-prompt SCST
    step with the reward taken through real strings, and the greedy baseline's decoded sections are returned for the caller to write back
    into its history (:243-246: they become the next study's prompt). The reference runs this at mbatch_size == 1 (:38) because consecutive
    studies of a patient depend on each other; any batch of INDEPENDENT studies works here.
    reward: reward.CXRBERTReward with a tokenizer. Returns scst_step's dict + "baseline_findings" / "baseline_impression" (lists of str)."""
    from .reward import ReportReward
    prompt = model.tokenize_prompt(previous_findings, previous_impression, tokenizer, decoder_max_len, add_bos_token_id=True)
    add = tokenizer.convert_tokens_to_ids("[PMT-SEP]")      # (the reference indexes additional_special_tokens_ids; same id, and this spelling survives transformers 5)
    special = dict(bos=tokenizer.bos_token_id, eos=tokenizer.eos_token_id, sep=tokenizer.sep_token_id, pad=tokenizer.pad_token_id, pmt_sep=add)
    labels = [[f"{i} {j}"] for i, j in zip(findings, impression)]                                   # :198
    rfn = ReportReward(model, tokenizer, reward, labels, tokenizer.bos_token_id, tokenizer.sep_token_id, tokenizer.eos_token_id)
    out = scst_step(model, opt, rfn, images, prompt["input_ids"], None, special, decoder_max_len, top_k=top_k, top_p=top_p, temperature=temperature,
                    reward_on_host=True)
    out["baseline_findings"], out["baseline_impression"] = rfn.last_sections                         # the second reward call decoded the greedy rows
    out["prompt_ids"] = prompt["input_ids"]
    return out
