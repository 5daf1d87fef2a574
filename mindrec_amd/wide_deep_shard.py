"""The row-sharded ("hybrid parallel") half of the Wide&Deep step: a mixin of WideDeepEngine (mindrec_amd/wide_deep.py).

Reference: README.md:140-144; models/wide_deep/train_and_eval_distribute.py:135-138 (gradients_mean=True); the reference's own
mechanism -- replicated ids, masked local Gather of the row slice, AllReduce of the [N, D] partials (wide_and_deep.py:232-249) --
has static shapes and no host round trip, which is what this protocol keeps while moving 1 / n of those bytes:

  requester                                   owner (= every rank, for its rows: id mod n, or hash(key) mod n)
  ---------                                   -----
  route_slots: position -> slot o * cap + j   .
  all-to-all  [n * cap] {id, weight}   --->   gather: ONE pass over the fused rows, one message row per slot
              (equal splits: static)          [looked-up row (16-bit / fp32) | wide weight * mask, 0 | pad]
  unroute_slots: message rows -> the    <---  all-to-all [n * cap, W]
     MLP input + the wide products            (side branch: Unique + inverted index of the received ids, under the MLP)
  MLP forward / backward (data parallel)
  route_grads: [row gradient | dlogit]  --->  all-to-all [n * cap, W] -> ONE apply kernel reads the message in place:
                                              LazyAdam on the deep columns + FTRL on the wide record of every touched row
  all-reduce(mean) of the dense gradient, dense Adam

Every rank hands every owner exactly `cap` = ceil(shard_capacity_factor * N / n) slots (unused ones carry id -1: the owner's
gather skips them, its plan sorts them behind the index proper, its apply never sees them), so every message has a static,
host-known shape: no bucket sizes cross the host, and with a capturable communicator (RCCL) the whole step -- collectives
included -- is ONE HIP graph per rank.  A position that finds its bucket full is dropped and counted in a sticky device
counter (`shard_overflow()`, checked by the caller once per sink: such a step is not a valid step).

UNIQUE-level exchange (`WideDeepConfig.shard_unique_factor` > 0; round 5).  The reference dedups in front of the sharded lookup
(Unique().shard(((1,),)), wide_and_deep.py:212; embedding.py:189-195), and Criteo-like batches hold 0.2-0.3 unique ids per position:
the requester routes its batch's UNIQUE ids (mrec_shard_route_slots_nv_*: the list's length lives on the device), the owner answers
one fp32 row per unique id ([row | wide weight], no mask), the requester fans the answers out to its positions with the one-GPU
fused lookup kernel over the returned message as its table (fp32 row x the position's weight, rounded once: bit-identical to one
GPU); backward the requester sums its positions' row gradients per unique id (mrec_segment_sum_g16, positions in ascending order)
and ships one fp32 sum per unique id, which the owner's one apply kernel reads in place.  Slots per owner: ceil(capacity_factor *
unique_factor * N / n) -- 336-byte fp32 rows against 176-byte 16-bit ones, so it pays below ~0.5 unique ids per position.

A rank's OWN chunk of a message never moves: the requester numbers the chunks of what it sends so that its own comes last
(owner o -> chunk (o - rank - 1) mod n: `chunk_rot` of mrec_shard_route_slots), an owner lays out what it receives so that its
own comes first (sender s -> chunk (s - rank) mod n), and the send buffer X[0 : n] and the receive buffer X[n - 1 : 2n - 1] of one
allocation then share exactly that chunk; the collective moves the other n - 1 (`_exchange`).  On one rank nothing is left to
move at all."""
import torch

from .wide_deep_mlp import _WideProd


class ShardCapacityError(RuntimeError):
    """A step's ids did not fit the fixed-capacity request message: raise WideDeepConfig.shard_capacity_factor."""


def grow_shard_capacity(eng, mult=2.0):
    """After a ShardCapacityError: the request messages get `mult` times the slots (the capacity factor, and the unique-ids-per-
    position bound up to 1) and the captured steps -- whose message shapes are part of them -- are dropped; the next step
    captures anew.  Every rank calls it (the error is raised on every rank alike).  The steps since the last check ran with
    zero rows for the dropped positions: the caller decides whether to restore a checkpoint or to go on."""
    cfg = eng.cfg
    uq = float(getattr(cfg, "shard_unique_factor", 0.0))
    if uq > 0.0 and uq < 1.0:
        cfg.shard_unique_factor = min(1.0, uq * mult)
    else:
        cfg.shard_capacity_factor = float(cfg.shard_capacity_factor) * mult
    eng.release_graphs()
    eng._guard.factor = cfg.shard_capacity_factor
    return cfg.shard_capacity_factor, float(getattr(cfg, "shard_unique_factor", 0.0))


class OverflowGuard:
    """Makes a dropped position impossible to miss: the library itself raises, on EVERY rank, at most one call late.

    The routing kernel counts positions that found their owner's bucket full in a sticky device counter.  Each step copies that
    counter into a spare word behind the dense gradient, so the all-reduce of the dense gradients -- which every step runs
    anyway -- hands every rank the SUM over all ranks (no extra collective: all ranks take the same branch, nobody is left
    inside a barrier).  After a step or sink the word is copied to pinned host memory asynchronously; the NEXT call into the
    engine (or close / checkpoint / an epoch's end, which wait for it) reads it and raises ShardCapacityError.  No host
    synchronisation is added to the step."""

    def __init__(self, slot, factor):
        self.slot, self.factor = slot, factor             # slot: one float32 element behind the dense gradient
        self.seen = 0.0
        self._host = torch.zeros(1, dtype=torch.float32).pin_memory() if slot.is_cuda else torch.zeros(1, dtype=torch.float32)
        self._event = None

    def stage(self, counter):
        """Inside the step (capturable): the local sticky counter -> the word the dense all-reduce will sum."""
        self.slot.copy_(counter)

    def probe(self):
        """Behind a step / sink, outside capture: start the copy of the summed word to the host."""
        if self.slot.is_cuda:
            self._host.copy_(self.slot, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host.copy_(self.slot)
            self._event = None

    def poll(self, block=False):
        """Raises if the last probed value shows new drops.  block=False: only if the copy has landed (no host wait)."""
        if self._event is not None:
            if block:
                self._event.synchronize()
            elif not self._event.query():
                return
            self._event = None
        n = float(self._host[0])
        if n > self.seen:
            d, self.seen = n - self.seen, n
            raise ShardCapacityError(f"{int(d)} id positions (all ranks) did not fit the fixed-capacity request message "
                                     f"(shard_capacity_factor {self.factor}): the steps since the last check are not valid "
                                     f"steps -- raise the factor")


class ShardStepMixin:
    def _shard_init(self):
        D = self.cfg.emb_dim
        # the own-chunk bypass needs a communicator that takes per-peer tensor lists (the product's; the CPU stand-in keeps the
        # plain equal-split exchange)
        self._bypass = bool((self._gpu or getattr(self, "_cpu_bypass", False)) and hasattr(self.comm, "all_to_all_lists"))
        if self._bypass and self.world > 1:
            self._bypass = self._exchange_selftest()
        self._act = self._amp if self._mfma else torch.float32           # dtype of looked-up rows / row gradients, also on the wire
        self._shard_fold = bool(self._fused_rows and D <= 252 and D % (4 if self._act == torch.float32 else 8) == 0)
        self._overflow = torch.zeros(1, dtype=torch.int64, device=self.device)
        n = self.dense_flat.numel()
        self._guard = OverflowGuard(self.dense_grad_full[n:n + 1], self.cfg.shard_capacity_factor)

    def shard_overflow(self):
        """Positions THIS rank dropped so far because an owner's bucket of the request message was full (host sync)."""
        return int(self._overflow.item())

    def check_shard_overflow(self):
        """Raises ShardCapacityError, on every rank alike, if any rank dropped positions since the last check (waits for the
        last step; train_step / train_steps / predict / release_graphs / save_checkpoint poll by themselves)."""
        self._guard.probe()
        self._guard.poll(block=True)

    # ---- the exchange ---------------------------------------------------------------------------------------------------
    def _xbuf(self, rows, width, dtype):
        """(first window, second window) of one exchange buffer for a message of `rows` = n * cap rows: what is laid out by
        rotated owner chunk goes into the first, what is laid out by sender into the second; they share the rank's own chunk."""
        n = self.world
        cap = rows // n
        X = torch.empty(((2 * n - 1) * cap, width), dtype=dtype, device=self.device)
        return X[: n * cap], X[(n - 1) * cap:]

    def _exchange(self, recv, send, to_owner):
        """All-to-all of the n - 1 chunks that belong to other ranks.  to_owner: requests / gradients (send laid out by rotated
        owner chunk, recv by sender); else answers (send laid out by sender, recv by rotated owner chunk)."""
        n, me = self.world, self.rank
        if n == 1:
            return
        cap = send.shape[0] // n
        ins, outs = [], []
        for r in range(n):
            if r == me:
                ins.append(send[:0])
                outs.append(recv[:0])
                continue
            by_owner, by_sender = (r - me - 1) % n, (r - me) % n
            a, b = (by_owner, by_sender) if to_owner else (by_sender, by_owner)
            ins.append(send[a * cap:(a + 1) * cap])
            outs.append(recv[b * cap:(b + 1) * cap])
        self.comm.all_to_all_lists(outs, ins)

    def _exchange_selftest(self):
        """One small exchange through `_exchange` at start-up, checked element by element; all ranks agree on the outcome (a
        communicator that refuses the call -- per-peer lists with an empty own entry -- or delivers the wrong chunks sends
        every rank to the plain equal-split all-to-all instead of failing in the first captured step; a refusal is an argument
        check of the library and comes on every rank alike, before anything is exchanged)."""
        n, me, cap = self.world, self.rank, 8
        ok = 1
        send, recv = self._xbuf(n * cap, 2, torch.float32)
        for r in range(n):                                   # chunk of owner r carries (me, r)
            c = (r - me - 1) % n
            send[c * cap:(c + 1) * cap, 0] = float(me)
            send[c * cap:(c + 1) * cap, 1] = float(r)
        try:
            self._exchange(recv, send, to_owner=True)
        except (TypeError, ValueError, NotImplementedError, RuntimeError) as e:
            # A refusal AT THE CALL: the library's argument checks (per-peer lists with an empty own entry, aliased windows) run
            # before anything is enqueued and -- same shapes, same code on every rank -- fail on every rank alike (torch raises
            # them as TypeError / ValueError, c10d's own checks as RuntimeError), so all ranks reach the all-reduce below
            # together.  What is NOT swallowed: anything surfacing later (the synchronize and the element checks below) -- an
            # asynchronous RCCL or device error is local to one rank, whose peers would be inside the all-to-all while it enters
            # an all-reduce.
            import warnings
            warnings.warn(f"own-chunk bypass refused by the communicator ({type(e).__name__}: {e}); using the equal-split all-to-all")
            ok = 0
        if ok:
            if self._gpu:
                torch.cuda.synchronize(self.device)
            for s_ in range(n):                                  # chunk of sender s must carry (s, me)
                c = (s_ - me) % n
                blk = recv[c * cap:(c + 1) * cap]
                ok &= int(bool((blk[:, 0] == float(s_)).all()) and bool((blk[:, 1] == float(me)).all()))
        flag = torch.tensor([float(ok)], device=self.device)
        self.comm.all_reduce(flag)
        return bool(flag.item() == float(n))

    # ---- forward half -------------------------------------------------------------------------------------------------
    def _answer(self, rows, rstride, wts, wstride, ns, out=None):
        """The owner's message [ns, W]: looked-up (masked) deep rows + the wide products of the requested rows."""
        cfg, k = self.cfg, self.k
        D, act = cfg.emb_dim, self._act
        if self._shard_fold:
            return k.gather_rows_req(self.deep, rows, rstride, wts, wstride, ns, D, act, out=out)
        # separate tables (split state, the cache tier, the CPU stand-in): two gathers, packed here
        assert rstride == 1 and wstride == 1
        Dw, W = k.shard_msg_words(D, act)
        msg = out if out is not None else torch.empty((ns, W), dtype=torch.float32, device=self.device)
        msg.zero_()
        e = k.gather_rows(self.deep, rows, wts) if act == torch.float32 else k.gather_rows(self.deep, rows, wts, out_dtype=act)
        msg[:, :Dw] = e.reshape(ns, D) if act == torch.float32 else e.reshape(ns, D).view(torch.float32)
        msg[:, Dw] = k.gather_rows(self.wide, rows, wts).reshape(ns)
        return msg

    def _shard_request(self, ids, wts):
        """The request half of a lookup: positions -> slots, the request exchange."""
        cfg, k = self.cfg, self.k
        n = ids.numel()
        cap = k.shard_capacity(n, self.world, cfg.shard_capacity_factor)
        if self._bypass:
            req, recv_req = self._xbuf(self.world * cap, 2, ids.dtype)
            _, slot_of_pos, pos_of_slot = k.shard_route_slots(ids, wts, self.world, cap, hashed=self._hashed, overflow=self._overflow,
                                                              rot=(self.rank + 1) % self.world, out=req)
            self._exchange(recv_req, req, to_owner=True)
        else:
            req, slot_of_pos, pos_of_slot = k.shard_route_slots(ids, wts, self.world, cap, hashed=self._hashed, overflow=self._overflow)
            recv_req = torch.empty_like(req)
            self.comm.all_to_all(recv_req, req)
        return {"recv_req": recv_req, "slot_of_pos": slot_of_pos, "pos_of_slot": pos_of_slot, "ns": self.world * cap}

    def _uniques_on(self):
        """Unique-level exchange: dense-storage fused rows under the 16-bit net (the benchmarked configuration)."""
        on = float(getattr(self.cfg, "shard_unique_factor", 0.0)) > 0.0
        if on and not (self._shard_fold and self._mfma and self.index is None and self.hb is None):
            raise ValueError("shard_unique_factor needs row shards of dense tables in the fused-row layout under the 16-bit net")
        return on

    def _shard_lookup_uniques(self, ids, wts, want_plan=True):
        """The lookup half with UNIQUE ids on the wire (module docstring).  Same return as _shard_lookup."""
        cfg, k = self.cfg, self.k
        B, Fd = ids.shape
        D, n = cfg.emb_dim, ids.numel()
        ev = self._tick("route")
        d = k.unique(ids)                                      # critical path: the request is made of the batch's unique ids
        capu = k.shard_capacity(max(int(n * float(cfg.shard_unique_factor)), 1), self.world, cfg.shard_capacity_factor)
        ns = self.world * capu
        if self._bypass:
            req, recv_req = self._xbuf(ns, 2, ids.dtype)
            _, slot_of_u, u_of_slot = k.shard_route_slots(d.uniq_buf, None, self.world, capu, overflow=self._overflow,
                                                          rot=(self.rank + 1) % self.world, out=req, n_valid_dev=d.n_uniq_dev)
            self._exchange(recv_req, req, to_owner=True)
        else:
            req, slot_of_u, u_of_slot = k.shard_route_slots(d.uniq_buf, None, self.world, capu, overflow=self._overflow,
                                                            n_valid_dev=d.n_uniq_dev)
            recv_req = torch.empty_like(req)
            self.comm.all_to_all(recv_req, req)
        self._tock(ev)
        ev = self._tick("gather_deep")
        W = k.shard_msg_words(D, torch.float32)[1]
        back = ans_out = None
        if self._bypass:
            back, ans_out = self._xbuf(ns, W, torch.float32)
        # the owner answers UNMASKED fp32 rows + the row's wide weight (weight 1 travelled in the request entries)
        if recv_req.dtype == torch.int32:
            ans = k.gather_rows_req(self.deep, recv_req, 2, recv_req.view(torch.float32).view(-1)[1:], 2, ns, D, torch.float32, out=ans_out)
        else:
            ans = k.gather_rows_req(self.deep, recv_req, 2, recv_req.view(torch.float32).view(-1)[2:], 4, ns, D, torch.float32, out=ans_out)
        fork_ev = None
        if self._side is not None and want_plan:
            fork_ev = torch.cuda.Event()
            fork_ev.record(torch.cuda.current_stream())
        self._tock(ev)
        ev = self._tick("a2a_rows")
        if self._bypass:
            self._exchange(back, ans, to_owner=False)
        else:
            back = torch.empty_like(ans)
            self.comm.all_to_all(back, ans)
        self._tock(ev)
        ev = self._tick("unroute")
        # fan-out: position p reads message row slot_of_u[inv[p]] -- the one-GPU fused lookup over the message as its table
        slot_of_pos = k.compose_i32(slot_of_u, d.inv)
        emb, wprod = k.gather_rows_wide(back[:, :D], slot_of_pos.view(B, Fd), wts, D, out=self._emb_out(n, D, self._amp), out_dtype=self._amp)
        self._tock(ev)
        plan_r = plan_o = rows = recv_wts = None
        if want_plan:
            with (torch.cuda.stream(self._side) if fork_ev is not None else _null()):
                if fork_ev is not None:
                    self._side.wait_event(fork_ev)
                plan_r = k.group_by_inverse(d)                  # the requester's inverted index: its backward sums per unique id
                rows, recv_wts = k.shard_unpack_req(recv_req)
                plan_o = k.sparse_plan(rows, skip_negative=True)     # the owner's: one entry per (rank, unique id) received
            if fork_ev is not None:
                main = torch.cuda.current_stream()
                for t in (rows, recv_wts, recv_req, plan_o.uniq_buf, plan_o.inv, plan_o.n_uniq_dev, plan_o.sorted_pos, plan_o.sorted_seg,
                          plan_o.seg_offsets, plan_r.sorted_pos, plan_r.sorted_seg, plan_r.seg_offsets, d.uniq_buf, d.inv):
                    self._rs(t, main)
                    self._rs(t, self._side)
        route = {"pos_of_slot": u_of_slot, "recv_wts": recv_wts, "plan": plan_o, "plan_r": plan_r, "ns": ns, "keep": (recv_req, rows, d),
                 "uniques": True}
        return emb.view(B, Fd * D), wprod.view(B, Fd, 2), route

    def _shard_lookup(self, ids, wts, want_plan=True):
        """Requests out, answers back.  Returns (emb [B, F * D] act dtype, wprod [B, F, 2] fp32, route state)."""
        if self._uniques_on():
            return self._shard_lookup_uniques(ids, wts, want_plan)
        cfg, k = self.cfg, self.k
        B, Fd = ids.shape
        D, n, act = cfg.emb_dim, ids.numel(), self._act
        ev = self._tick("route")
        rq = self._shard_request(ids, wts)
        recv_req, slot_of_pos, pos_of_slot, ns = rq["recv_req"], rq["slot_of_pos"], rq["pos_of_slot"], rq["ns"]
        self._tock(ev)
        ev = self._tick("gather_deep")
        plan = recv_wts = rows = None
        back = ans_out = None
        if self._bypass:
            back, ans_out = self._xbuf(ns, k.shard_msg_words(D, act)[1], torch.float32)
        translate = self.index is not None or self.hb is not None
        if translate:
            # hash tables / the cache tier: what arrived are raw keys (or rows of a host table); the owner's own index gives them
            # rows -- new keys: the next rows, default values keyed by the key; padding (-1) stays -1
            recv_ids, recv_wts = k.shard_unpack_req(recv_req)
            if self.hb is not None:
                plan, rows = self.hb.prepare(recv_ids, skip_negative=True)
            else:
                rows = self.index.lookup(recv_ids, insert=True, tables=self._map_tables(), skip_pad=True)
            ans = self._answer(rows, 1, recv_wts, 1, ns, out=ans_out)
        elif self._shard_fold:
            # ids and weights are read straight out of the received entries ({id, weight}: stride 2 in units of either)
            if recv_req.dtype == torch.int32:
                ans = self._answer(recv_req, 2, recv_req.view(torch.float32).view(-1)[1:], 2, ns, out=ans_out)
            else:
                ans = self._answer(recv_req, 2, recv_req.view(torch.float32).view(-1)[2:], 4, ns, out=ans_out)
        else:
            rows, recv_wts = k.shard_unpack_req(recv_req)
            ans = self._answer(rows, 1, recv_wts, 1, ns, out=ans_out)
        fork_ev = None
        if self._side is not None and want_plan:
            # the step's Unique + inverted index of the received ids: a dozen small latency-bound kernels, on the side branch
            # under the MLP.  Marked here, ISSUED behind the answer exchange and the un-permute (under capture the branch whose
            # first node is created first stays on the launch queue: that must be the critical chain)
            fork_ev = torch.cuda.Event()
            fork_ev.record(torch.cuda.current_stream())
        self._tock(ev)
        ev = self._tick("a2a_rows")
        if self._bypass:
            self._exchange(back, ans, to_owner=False)
        else:
            back = torch.empty_like(ans)
            self.comm.all_to_all(back, ans)
        self._tock(ev)
        ev = self._tick("unroute")
        eo = self._emb_out(n, D, act) if act != torch.float32 else None       # static graph input, when the MLP graph exists
        emb, wprod = k.shard_unroute_slots(back, slot_of_pos, D, act, out=eo)
        self._tock(ev)
        if want_plan:
            with (torch.cuda.stream(self._side) if fork_ev is not None else _null()):
                if fork_ev is not None:
                    self._side.wait_event(fork_ev)
                if rows is None:
                    rows, recv_wts = k.shard_unpack_req(recv_req)
                if plan is None:
                    plan = k.sparse_plan(rows, skip_negative=True)
            if fork_ev is not None:
                main = torch.cuda.current_stream()
                for t in (rows, recv_wts, recv_req, plan.uniq_buf, plan.inv, plan.n_uniq_dev, plan.sorted_pos, plan.sorted_seg, plan.seg_offsets):
                    self._rs(t, main)
                    self._rs(t, self._side)
        # (recv_req is read by the side branch: it must stay alive until the branches join -- under capture a freed block is
        # handed to the next allocation of the SAME capture, whatever another branch still does with it)
        route = {"pos_of_slot": pos_of_slot, "recv_wts": recv_wts, "plan": plan, "ns": ns, "keep": (recv_req, rows)}
        return emb.view(B, Fd * D), wprod.view(B, Fd, 2), route

    def _front_sharded(self, ids, wts, label, capturing=False):
        cfg = self.cfg
        emb, wprod, route = self._shard_lookup(ids, wts)
        ev = self._tick("mlp_fwd_bwd")
        fused = self._mfma
        if fused:
            K5 = self.dims[len(self.dims) - 2]
            if self.k.head_supported(K5):
                wide = _WideProd(wprod)            # the per-sample sum over the fields (+ Wide_b) is taken inside the output head
            else:
                wide = wprod[..., 0].sum(dim=1) + self.wide_b
            step = self._mlp_step_eager if capturing else self._mlp_step
            loss, g_emb, g_wide = step(emb, wide, label)
            route["wide_b_in_head"] = isinstance(wide, _WideProd)
        elif self._f32net:
            loss, g_emb, g_wide = self._mlp_step_f32(emb, wprod[..., 0].sum(dim=1) + self.wide_b, label)
            route["wide_b_in_head"] = False
        else:
            loss, g_emb, g_wide = self._mlp_step_generic(emb, wprod[..., 0].sum(dim=1) + self.wide_b, label)
            route["wide_b_in_head"] = False
        self._tock(ev)
        return loss, g_emb, g_wide, route["plan"], True, route, None, fused

    # ---- optimizer half -----------------------------------------------------------------------------------------------
    def _tail_sharded(self, front, ids, wts):
        cfg, k = self.cfg, self.k
        B, Fd = ids.shape
        D, n, act = cfg.emb_dim, ids.numel(), self._act
        loss, g_emb, g_wide, plan, _, route, _, fused = front
        inv_sens = 1.0 / cfg.sens
        state = None
        if self._dyn:
            state = self._step_state
            state.advance(cfg.adam_lr, float(self.beta1), float(self.beta2))
        ev = self._tick("a2a_grads")
        uniques = bool(route.get("uniques"))
        if uniques:
            # one fp32 SUM per unique id: this rank's positions summed in ascending order (the Mul bprops of the masks as row_scale),
            # the wide branch's per-position gradient dlogit[sample] * weight likewise
            act = torch.float32
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)       # the requester's inverted index (queued under the MLP)
            sums = k.segment_sum(route["plan_r"], g_emb.reshape(n, D), wts)
            gw_u = k.segment_sum(route["plan_r"], g_wide.view(B, 1).expand(B, Fd).reshape(n, 1), wts).reshape(-1)
            g_rows, g_dl, g_F = sums, gw_u, 1
        else:
            g_rows, g_dl, g_F = g_emb.reshape(n, D), g_wide.reshape(B).contiguous(), Fd
        if self._bypass:
            gmsg, recv_g = self._xbuf(route["ns"], k.shard_msg_words(D, act)[1], torch.float32)
            k.shard_route_grads(g_rows, g_dl, g_F, route["pos_of_slot"], out=gmsg)
            self._exchange(recv_g, gmsg, to_owner=True)
        else:
            gmsg = k.shard_route_grads(g_rows, g_dl, g_F, route["pos_of_slot"])
            recv_g = torch.empty_like(gmsg)
            self.comm.all_to_all(recv_g, gmsg)
        self._tock(ev)
        # Dense gradients (+ Wide_b's, an element of the same buffer): all-reduce queued behind the row-gradient exchange and
        # left running (RCCL's stream) while the sparse apply executes -- it does not need it.  (The slab sum + all-reduce on a
        # side branch of their own, and the NEXT step's request exchange issued under this step's apply inside a sink's graph,
        # were both measured on one box: 0.905 ms/step as it is, 0.963 / 1.003 / 0.946 with either or both -- every extra branch
        # costs this graph runtime more than the overlap returns.)
        ev = self._tick("allreduce_dense")
        if fused:
            self._sum_dw_slabs()
            if not route["wide_b_in_head"]:
                self.wide_b_grad.copy_(self.dense_grad[2 * (len(self.dims) - 2) + 1].view(1))      # = the output layer's bias gradient
        else:
            self.wide_b_grad.copy_(g_wide.sum().view(1))
        self._guard.stage(self._overflow)                                  # the dropped-position count rides the dense all-reduce
        dense_work = self.comm.all_reduce(self.dense_grad_full, async_op=True)
        self._tock(ev)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)           # the plan (queued under the MLP) is done
        # RowTensor gradients of all ranks are summed at the owner; gradients_mean divides by the number of ranks
        scale = inv_sens / self.world
        Dw, _ = k.shard_msg_words(D, act)
        recv_rows = recv_g[:, :D] if act == torch.float32 else recv_g.view(act)[:, :D]       # [ns, D], read in place
        recv_gw = recv_g[:, Dw:Dw + 1]                                                         # [ns, 1] fp32: one value per position
        akw = dict(lr=cfg.adam_lr, beta1=float(self.beta1), beta2=float(self.beta2), eps=cfg.adam_eps,
                   beta1_power=float(self.beta1_power), beta2_power=float(self.beta2_power), grad_scale=scale)
        ev = self._tick("apply_deep")
        if self.deep_apply_timer is not None:
            self.deep_apply_timer.arm()
            self.deep_apply_timer = None
        if self._shard_fold:
            # LazyAdam + the wide record's FTRL in one visit per touched row, as on one GPU; the wide gradient of a received
            # position is a column of the gradient message (one value per position: F = 1)
            k.sparse_lazy_adam_wide_(self.deep, self.deep_m, self.deep_v, plan, recv_rows, route["recv_wts"], recv_gw, 1, D,
                                     ftrl_lr=cfg.ftrl_lr, l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, step_state=state, **akw)
        else:
            k.sparse_lazy_adam_(self.deep, self.deep_m, self.deep_v, plan, recv_rows, route["recv_wts"], **akw)
            k.sparse_ftrl_(self.wide, self.wide_accum, self.wide_linear, plan, recv_gw, route["recv_wts"], lr=cfg.ftrl_lr,
                           l1=cfg.ftrl_l1, l2=cfg.ftrl_l2, grad_scale=scale)
        self._tock(ev)
        if dense_work is not None:
            dense_work.wait()                     # the current stream waits for RCCL's stream; no host block
        ev = self._tick("apply_dense")
        flat = self.dense_flat.detach()
        if fused:
            # (the weight-gradient slabs were summed for the all-reduce above; the kernel also refreshes the 16-bit operand shadow)
            k.dense_adam_slabs_(flat, self.dense_m, self.dense_v, self.dense_grad_flat, [], shadow16=self.dense16_flat,
                                step_state=state, ftrl1=self._wb_ftrl, **akw)
            self._refresh_tail()
        else:
            k.dense_adam_(flat, self.dense_m, self.dense_v, self.dense_grad_flat, ftrl1=self._wb_ftrl, **akw)
        self._tock(ev)
        self.last_plan = plan
        return loss.detach()


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
