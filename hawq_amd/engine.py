"""Fused integer executor for a frozen Q_ResNet on one MI355X.

Walks the frozen network once on the host (the reference redoes all of this on every forward:
quant_modules.py:441-484, quant_utils.py:188-213), producing
  * packed int8 / hawq4 weights, int32 biases, per-channel dyadic (m, e) tables on the device,
  * a launch list of C-ABI calls (include/hawq_mi355.h) in which every tensor between kernels
    is an integer NHWC tensor at its schedule bit-width:

      quantize_input -> stem_conv7(+QuantAct16+ReLU) -> maxpool(+unit1 QuantAct)
      per unit:  conv1(+ReLU+QuantAct) -> [conv2(+ReLU+QuantAct)] ->
                 conv_last(+identity conv in the same launch | +16-bit residual read)
                          (+residual add, ReLU, next unit's QuantAct, 16-bit residual out)
      avgpool(+QuantAct8) -> FC(+per-class dequant) -> fp32 logits

  * optionally a hipGraph of the whole list, replayed per batch.

Integer semantics follow SURVEY.md App. A; the graph is q_resnet.py:53-74 / 114-135 / 231-316.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from functools import partial

import numpy as np
import torch

from . import _lib, packing
from .quant_utils import requant_table, tables_are_fast, tables_fit_fast


RES_VBITS = 20  # residual / pooled values are 16-bit-ish (uint16 storage saturates at 65535)
U16_VBITS = 17  # tie-freeness proofs for values that live in uint16 residual tensors (< 2^16 unless flagged)


def _i32(arr, dev):
    return torch.from_numpy(np.ascontiguousarray(arr, np.int32)).to(dev)


def _no_preshift(ek) -> bool:
    """True if no entry of a device table (e | k << 8) carries a pre-shift k."""
    return bool((np.asarray(ek, dtype=np.int64) >> 8 == 0).all())


def _act_range(bits, mode):
    if mode == 'symmetric':
        return -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    return 0, 2 ** bits - 1


class _OpList(list):
    """launch list; ``names[i]`` labels ``self[i]`` (set ``next_name`` before appending)."""

    def __init__(self):
        super().__init__()
        self.names = []
        self.next_name = None

    def append(self, fn):
        name = self.next_name or (fn.args[0] if hasattr(fn, "args") else "op")
        self.names.append(name)
        self.next_name = None
        super().append(fn)


class StalePlan(ValueError):
    """A recorded plan does not fit the launch list / kernel inventory of this build: the engine falls back to tuning."""


class _FusedPair:
    """One entry of the launch list that covers TWO layers - the expand conv (+ residual epilogue) of unit i and the reduce
    conv of unit i+1 - either as one hawq_conv_expand_reduce launch or as two hawq_conv2d launches, whichever the
    autotuner measured faster for this batch shape (results are identical)."""

    def __init__(self, er, expand, reduce, stream):
        # reduce None: a "solo" entry - the expand conv alone, either through hawq_conv_expand_reduce (reduce.wgt == NULL:
        # the wave-private kernel of fused_wp.hip) or through hawq_conv2d
        self.er, self.expand, self.reduce, self.sp = er, expand, reduce, stream
        self.fused = True

    def __call__(self):
        if self.fused:
            _lib.call("hawq_conv_expand_reduce", C.byref(self.er), self.sp)
        else:
            _lib.call("hawq_conv2d", C.byref(self.expand), self.sp)
            if self.reduce is not None:
                _lib.call("hawq_conv2d", C.byref(self.reduce), self.sp)


class _Conv:
    """Device-resident parameters of one QuantBnConv2d."""

    def __init__(self, mod, s_a, in_bits, dev, from_buffers, in_range=None):
        if not from_buffers:
            mod.prepare(s_a)
        w_int = mod.weight_integer.detach().cpu().numpy()
        self.cout, self.cin, self.kh, self.kw = w_int.shape
        self.stride, self.pad = int(mod.conv.stride[0]), int(mod.conv.padding[0])
        if mod.conv.groups != 1 or mod.conv.dilation[0] != 1:
            raise NotImplementedError("grouped/dilated convolutions are outside the ResNet hot path")
        # storage of the weights follows the storage of the activations they meet (see IntegerEngine._storage): nibbles only
        # for 4-bit x 4-bit layers, int8 otherwise - a 4-bit weight in an int8 byte is the same integer
        self.w_bits = 4 if (mod.weight_bit <= 4 and in_bits == 4) else 8
        self.in_bits = in_bits
        self.band_ok = self.k128_ok = False
        self.s_w = mod.convbn_scaling_factor.detach().float().cpu()
        self.w_host = w_int
        b = mod.bias_integer.detach().cpu().numpy().astype(np.float64)
        self.b_host = np.clip(np.rint(b), -2 ** 31, 2 ** 31 - 1).astype(np.int64)
        if self.cin % 64 == 0 and self.cout % 64 == 0:
            wp = packing.pack_conv_weight(w_int, self.w_bits)
            self.w = torch.from_numpy(wp).to(dev)
            # The round-5 kernels stream the same integers in their own tile order (include/hawq_mi355.h: hawq_conv_args.wgt_band /
            # wgt_k128 / wgt2_k128).  Those copies are packed on first use (band() / k128()) and dropped again by
            # IntegerEngine._release_unused_weights() for layers whose chosen tile is not such a kernel: a plan that never picks them
            # does not hold the network's weights three times.
            self._wp, self._dev = wp, dev
            rowb = self.cin * self.w_bits // 8   # bytes of a filter tap (hawq4: two channels per byte, Cin % 128 == 0 for the nibble kernels)
            self.band_ok = ((self.kh, self.kw, self.stride, self.pad) == (3, 3, 1, 1) and rowb >= 128 and rowb % 64 == 0
                            and (self.w_bits == 8 or self.cin % 128 == 0) and not os.environ.get("HAWQ_NO_BAND2"))
            # 128-byte K chunks: int8 rows of Cin % 128 == 0 channels, or (round 6) hawq4 rows of Cin % 256 == 0 channels
            self.k128_ok = ((self.kh, self.kw, self.pad) == (1, 1, 0) and rowb % 128 == 0 and not os.environ.get("HAWQ_NO_GEMM2")
                            and (self.w_bits == 8 or not os.environ.get("HAWQ_NO_GEMM2_NIB")))
        self._w_band = self._w_k128 = None
        self.bias = _i32(self.b_host, dev)
        # exact per-channel bound on |accumulator| -> bit length, for the requant pre-shift check
        amax = max(abs(int(in_range[0])), abs(int(in_range[1]))) if in_range is not None else (128 if in_bits == 8 else 15)
        bound = np.abs(np.rint(w_int.astype(np.float64))).reshape(self.cout, -1).sum(1) * amax + np.abs(self.b_host)
        self.vbits = np.array([int(b).bit_length() for b in bound], np.int64)
        if (self.vbits > 31).any():
            raise ValueError("int32 accumulator overflow is possible for this layer")

    def band(self):
        """hawq_pack_w3x3_band stream of this layer's weights on the device (None if no round-5 3x3 kernel could take it)."""
        if self.band_ok and self._w_band is None:
            self._w_band = torch.from_numpy(packing.pack_w3x3_band(self._wp, self.cout, self.cin * self.w_bits // 8)).to(self._dev)
        return self._w_band

    def k128(self):
        """hawq_pack_w1x1_k128 stream (128-byte K chunks) of this layer's weights on the device, or None."""
        if self.k128_ok and self._w_k128 is None:
            self._w_k128 = torch.from_numpy(packing.pack_w1x1_k128(self._wp, self.cout, self.cin * self.w_bits // 8)).to(self._dev)
        return self._w_k128

    def release(self, band: bool, k128: bool):
        if band:
            self._w_band = None
        if k128:
            self._w_k128 = None


class IntegerEngine:
    """Callable: fp32 NCHW images on the GPU -> fp32 logits, bit-identical to the reference's
    frozen forward.  ``residual_bits`` 16 stores post-ReLU residuals as uint16 with a sticky
    overflow flag (``overflowed()``); 32 stores int32.  ``from_buffers`` trusts the modules'
    integer buffers / scales as loaded from a quantized checkpoint (quant_train.py:665-670)
    instead of re-deriving them from the float parameters."""

    # launches that recorded plans may still list as tuned although this build runs them on a kernel with nothing to tune
    _UNTUNED_SINCE = ("quant_output",)

    def __init__(self, model, residual_bits: int = 16, from_buffers: bool = False, use_graph: bool = True,
                 keep_accumulators: bool = False, fast: bool = True, autotune: bool = True, chains: int = 0,
                 plan=None, _parent=None):
        if _parent is not None:  # a chain of a multi-chain engine: shares parameters, owns stream + buffers
            self.__dict__.update({k: v for k, v in _parent.__dict__.items()
                                  if k in ("model", "dev", "res_bits", "from_buffers", "keep_acc", "fast", "autotune",
                                           "flags", "P", "planar", "fuse_stages", "plan")})
            self._plan_on = False
            self.use_graph, self.chains, self.subs = False, 1, []
            self.stream = torch.cuda.Stream(device=self.dev)
            self.tile_choice, self.er_choice = {}, {}
            self._batch = self._graph = None
            return
        if not model.is_frozen():
            raise RuntimeError("IntegerEngine needs a frozen model (freeze_model) - ranges must be fixed")
        _lib.load()
        self.model = model
        self.dev = next(model.parameters()).device
        if self.dev.type != 'cuda':
            raise RuntimeError("IntegerEngine: move the model to the MI355X first (no CPU path)")
        self.res_bits = residual_bits
        self.from_buffers = from_buffers
        self.use_graph = use_graph and not os.environ.get("HAWQ_NO_GRAPH")  # (HAWQ_NO_GRAPH: direct launches, for experiments)
        self.keep_acc = keep_accumulators
        self.fast = fast  # False forces the exact general kernels everywhere (reference for tests)
        self.autotune = autotune  # pick each conv launch's tile configuration by timing it once per batch shape
        # conv1 -> 3x3 conv2 tensors as channel-group planes (include/hawq_mi355.h: in_planar / out_planar)
        self.planar = not keep_accumulators and not os.environ.get("HAWQ_NO_PLANAR")
        # expand conv of unit i + reduce conv of unit i+1 in one launch (hawq_conv_expand_reduce) in these stages
        self.fuse_stages = set() if keep_accumulators else {int(v) for v in os.environ.get("HAWQ_FUSE_STAGES", "1,2,3").split(",") if v}
        self.tile_choice = {}
        self.er_choice = {}
        # chains > 1: the batch is split into independent sub-batches whose launch chains run on separate
        # streams inside ONE hipGraph, so that one chain's kernel tails / prologues / epilogues overlap the
        # other's work (the late layers launch fewer workgroups than there are CUs).  0 = choose 1, 2 or 3 by
        # timing the captured graph once per batch shape (HAWQ_CHAINS overrides).
        self.chains_req = 1 if keep_accumulators else max(0, int(os.environ.get("HAWQ_CHAINS", chains)))
        self.chains = max(1, self.chains_req)
        # a recorded plan (export_plan(): chain count, tile id of every conv launch, fused variant of every expand(-> reduce) launch)
        # for ONE batch shape: that shape is built by replaying it - no timing at all -, any other shape is tuned as usual.  What
        # bench.py --plan and the multi-GPU path (rank 0 tunes, every rank replays: hawq_amd.dist.share_plan) hand over.
        self.plan = dict(plan) if plan else None
        # storage policy of 4-bit expand-conv inputs (see _prepare_params): "0" nibbles, "1" int8 in a fusable unit whose successor's block
        # input is 8-bit (rounds 3-5), "2" int8 in every identity-pass-through bottleneck (round 6, the default; env HAWQ_EXPAND_IN8).
        # A recorded plan carries the policy its launch list was built with.
        # (a plan recorded before round 6 has no such key: it was recorded under policy "1")
        self.expand_in8 = str(self.plan.get("expand_in8") or "1") if self.plan else os.environ.get("HAWQ_EXPAND_IN8", "2")
        self.plan_source = "tuned in this process"
        self.subs = []
        self.stream = torch.cuda.Stream(device=self.dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._batch = None
        self._graph = None
        # uint16 residual plan: a forward whose un-clamped residual (quant_utils.py:456) exceeds 65535 raises the
        # sticky device flag; the public entry points then redo that batch on a cached int32-residual twin
        self._wide = None
        self.overflow_fallbacks = 0
        self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._prepare_params()

    # ------------------------------------------------------------------ host-side preparation
    def _scale(self, act):
        if self.from_buffers:
            return act.act_scaling_factor.detach().float().cpu().reshape(-1)[:1]
        return act.compute_scale().detach().float().cpu().reshape(-1)[:1]

    def _prepare_params(self):
        m, dev = self.model, self.dev
        one = torch.ones(1)
        P = self.P = {}
        qi = m.quant_input
        if qi.activation_bit != 8 or qi.quant_mode != 'symmetric':
            raise NotImplementedError("quant_input must be 8-bit symmetric (every shipped schedule)")
        s_in = self._scale(qi)
        P['s_in'] = s_in
        P['inv_s_in'] = float((1. / s_in).item())
        stem = m.stem
        if tuple(stem.conv.kernel_size) != (7, 7) or stem.conv.stride[0] != 2 or stem.conv.in_channels > 4:
            raise NotImplementedError("stem must be the 7x7/2 conv of the ImageNet ResNets")
        sc = _Conv(stem, s_in, 8, dev, self.from_buffers, _act_range(qi.activation_bit, qi.quant_mode))
        sc.w = torch.from_numpy(packing.pack_stem_weight(sc.w_host)).to(dev)
        a0 = m.quant_act_int32
        s0 = self._scale(a0)
        mm, ee = requant_table(s_in, sc.s_w, s0, vbits=sc.vbits)
        P['stem'] = dict(conv=sc, m=_i32(mm, dev), e=_i32(ee, dev), rng=_act_range(a0.activation_bit, a0.quant_mode),
                         fast=tables_are_fast(mm, ee, sc.vbits))
        s_prev = s0
        units = []
        unit_list = list(m.units())

        def block_input_bits(unit):
            return self._storage(self._store_bits(unit.quant_act), [unit.quant_convbn1] + ([unit.quant_identity_convbn] if unit.resize_identity else []))

        for ui, (name, u) in enumerate(unit_list):
            nxt_u = unit_list[ui + 1][1] if ui + 1 < len(unit_list) else None
            d = dict(name=name, resize=bool(u.resize_identity), nb=u.n_body)
            qa = u.quant_act
            s_a = self._scale(qa)
            d['a_bits'] = self._storage(self._store_bits(qa), [u.quant_convbn1] + ([u.quant_identity_convbn] if u.resize_identity else []))
            d['a_rng'] = _act_range(qa.activation_bit, qa.quant_mode)
            mq, eq = requant_table(s_prev, one, s_a, vbits=RES_VBITS)
            d['mq'], d['eq'] = int(mq[0]), int(eq[0])
            if not units:  # the stem kernel applies the first unit's QuantAct
                P['stem']['fast'] = P['stem']['fast'] and tables_are_fast(mq, eq, U16_VBITS)
            if d['resize']:
                d['ident'] = _Conv(u.quant_identity_convbn, s_a, d['a_bits'], dev, self.from_buffers, d['a_rng'])
            s_x, bits_x, rng_x = s_a, d['a_bits'], d['a_rng']
            d['convs'] = []
            for i in range(1, u.n_body + 1):
                c = _Conv(getattr(u, f"quant_convbn{i}"), s_x, bits_x, dev, self.from_buffers, rng_x)
                ent = dict(conv=c)
                if i < u.n_body:
                    act = getattr(u, f"quant_act{i}")
                    s_n = self._scale(act)
                    mm, ee = requant_table(s_x, c.s_w, s_n, vbits=c.vbits)
                    store = self._storage(self._store_bits(act), [getattr(u, f"quant_convbn{i + 1}")])
                    fusable = (int(name.split('.')[0][len('stage'):]) in self.fuse_stages and nxt_u is not None and not nxt_u.resize_identity)
                    # (a resize unit's expand conv shares its launch with the identity conv, whose input - the block input - keeps its own
                    #  storage: the dual launch wants equal operand widths in both branches, its exact-tie instantiations exist for nothing else)
                    if (store == 4 and u.n_body == 3 and i + 1 == u.n_body
                            and ((self.expand_in8 == "2" and not u.resize_identity) or (self.expand_in8 == "1" and fusable and block_input_bits(nxt_u) == 8))):
                        # Mixed schedules (8-bit block inputs, 4-bit tensors inside the units): the 4-bit input of an expand conv
                        # whose successor's reduce conv runs the int8 pipeline anyway is stored as int8, so that the fused
                        # expand -> reduce launch takes the pair (it packs the reduce conv's 4-bit output itself).  Measured
                        # (tools/nibble_pairs_ab.sh): bops_0.5 +1.6 %; pure W4A4 - whose block inputs are nibbles too - LOST
                        # 1.3 % in round 3 with its pairs fused on int8 operands and kept its nibble launches (policy "1", the default of rounds 3-5).
                        # Policy "2" (round 6, the default; recorded per plan as `expand_in8`): the 4-bit expand input of EVERY identity-pass-through
                        # bottleneck is stored as int8 - the expand(-> reduce) launches of a 4-bit schedule are then exactly the W8A8 plan's (same fused
                        # pairs, same wave-private solo launches; they are bound by the residual epilogue, not by their K = 64..512 bytes)
                        # while its 3x3 convs, un-fused reduce convs and identity convs stream nibbles (band_v2 / gemm_v2 NIB).  Same box,
                        # fresh plans (profiles/r06_w4a4_ab.txt): W4A4 98.7-98.9 k (policy 1) -> 99.7 k img/s, W8A8 96.4-97.9 k.
                        store = 8
                    ent.update(m=_i32(mm, dev), e=_i32(ee, dev), out_bits=store,
                               rng=_act_range(act.activation_bit, act.quant_mode),
                               fast=tables_fit_fast(mm, ee, c.vbits), tie=not tables_are_fast(mm, ee, c.vbits),
                               k0=_no_preshift(ee), ck0=_no_preshift(ee))
                    if ent['fast']:
                        ent['ctab'] = _i32(packing.pack_ctab(c.b_host, mm, ee), dev)
                    s_x, bits_x, rng_x = s_n, ent['out_bits'], ent['rng']
                else:
                    ent['s_last'] = s_x
                d['convs'].append(ent)
            ao = u.quant_act_int32
            s_o = self._scale(ao)
            last = d['convs'][-1]
            mm, ee = requant_table(last['s_last'], last['conv'].s_w, s_o, vbits=last['conv'].vbits)
            last.update(m=_i32(mm, dev), e=_i32(ee, dev), k0=_no_preshift(ee), ck0=_no_preshift(ee))
            fast = tables_fit_fast(mm, ee, last['conv'].vbits)
            tie = not tables_are_fast(mm, ee, last['conv'].vbits)
            if fast:
                last['ctab'] = _i32(packing.pack_ctab(last['conv'].b_host, mm, ee), dev)
            if d['resize']:
                m1, e1 = requant_table(s_a, d['ident'].s_w, s_o, vbits=d['ident'].vbits)
                d['m_id'], d['e_id'] = _i32(m1, dev), _i32(e1, dev)
                last['k0'] = last['k0'] and _no_preshift(e1)
                last['ck0'] = last['ck0'] and _no_preshift(e1)
                if tables_fit_fast(m1, e1, d['ident'].vbits):
                    d['ctab_id'] = _i32(packing.pack_ctab(d['ident'].b_host, m1, e1), dev)
                    tie = tie or not tables_are_fast(m1, e1, d['ident'].vbits)
                else:
                    fast = False
            else:
                m1, e1 = requant_table(s_prev, one, s_o, vbits=RES_VBITS)
                d['m_id_s'], d['e_id_s'] = int(m1[0]), int(e1[0])
                fast = fast and tables_fit_fast(m1, e1, U16_VBITS, allow_shift=True)
                tie = tie or not tables_are_fast(m1, e1, U16_VBITS, allow_shift=True)
            last['fast'], last['tie'] = fast, tie
            # the next unit's block-input QuantAct is fused into this launch: its table must fit the fast path too
            if units:
                prev_last = units[-1]['convs'][-1]
                prev_last['fast'] = prev_last['fast'] and tables_fit_fast([d['mq']], [d['eq']], U16_VBITS)
                prev_last['tie'] = prev_last['tie'] or not tables_are_fast([d['mq']], [d['eq']], U16_VBITS)
                prev_last['k0'] = prev_last['k0'] and _no_preshift([d['eq']])
            s_prev = s_o
            units.append(d)
        P['units'] = units
        ao = m.quant_act_output
        s8 = self._scale(ao)
        mq, eq = requant_table(s_prev, one, s8, vbits=RES_VBITS)
        P['out'] = dict(mq=int(mq[0]), eq=int(eq[0]), rng=_act_range(ao.activation_bit, ao.quant_mode))
        fc = m.quant_output
        if not self.from_buffers:
            fc.prepare(s8)
        w = fc.weight_integer.detach().cpu().numpy()
        nout, k = w.shape
        if k % 64:
            raise NotImplementedError("FC input features must be a multiple of 64")
        nout_p = (nout + 63) // 64 * 64
        s_fc = fc.fc_scaling_factor.detach().float().cpu()
        fscale = np.zeros(nout_p, np.float32)
        fscale[:nout] = (s_fc.view(1, -1) * s8.view(1, -1)).numpy().reshape(-1)  # fl(S_fc[c]*S_a)
        b = np.zeros(nout_p, np.int64)
        b[:nout] = np.clip(np.rint(fc.bias_integer.detach().cpu().numpy().astype(np.float64)), -2 ** 31, 2 ** 31 - 1)
        P['fc'] = dict(w=torch.from_numpy(packing.pack_conv_weight(w.reshape(nout, k, 1, 1), 8, k, nout_p)).to(dev),
                       bias=_i32(b, dev), fscale=torch.from_numpy(fscale).to(dev), nout=nout, nout_p=nout_p, k=k)

    @staticmethod
    def _storage(value_bits, consumers):
        """Storage width of an activation tensor whose VALUES are `value_bits` wide: hawq4 nibbles only if every conv that
        reads it is a 4-bit x 4-bit layer the nibble pipelines take (Cin % 128 == 0: a 64-byte LDS row is 128 channels);
        int8 otherwise.  Mixed-width layers (W4A8 / W8A4 of the latency_* / modelsize_* schedules) and the 64-channel
        4-bit layers of stage 1 then run the int8 asynchronous pipelines, band kernels and fused launches instead of the
        register-staged loop; the integers - hence every result - are the same, the tensor is at most 2x the bytes.
        HAWQ_NATIVE_STORAGE=1 restores one nibble per 4-bit value everywhere (A/B measurements)."""
        if value_bits != 4 or os.environ.get("HAWQ_NATIVE_STORAGE"):
            return value_bits
        ok = all(m.weight_bit <= 4 and m.conv.in_channels % 128 == 0 and m.conv.groups == 1 for m in consumers)
        return 4 if ok else 8

    @staticmethod
    def _store_bits(act):
        if act.activation_bit <= 4:
            if act.quant_mode != 'asymmetric':
                raise NotImplementedError("4-bit activations are stored unsigned (asymmetric mode)")
            return 4
        if act.activation_bit <= 8:
            # stored in int8 operands: an 8-bit 'asymmetric' range (0..255) would wrap, narrower ones fit
            if _act_range(act.activation_bit, act.quant_mode)[1] > 127:
                raise NotImplementedError("8-bit activations must be 'symmetric' (an unsigned 0..255 range does not fit "
                                          "the int8 conv operands)")
            return 8
        raise NotImplementedError("conv inputs wider than 8 bits")

    # ------------------------------------------------------------------ launch list
    def _band_takes(self, ent, N, h, w, in_bits, is_last, u) -> bool:
        """Would a 3x3 band kernel take the conv `ent` fed with an [N,h,w] map of `in_bits`-bit activations?  Asked
        of the library itself (hawq_conv2d_band_tile) with the launch's geometry, widths and epilogue class."""
        c = ent['conv']
        if not ent.get('fast', False) or self.res_bits != 16 or not self.fast:
            return False
        q = _lib.ConvArgs()
        q.N, q.H, q.W, q.Cin, q.Cout, q.KH, q.KW, q.stride, q.pad = N, h, w, c.cin, c.cout, c.kh, c.kw, c.stride, c.pad
        q.in_bits, q.w_bits, q.fast_tables = in_bits, c.w_bits, 1
        if is_last:  # RESIDUAL epilogue: single-branch uint16 residual only
            if u['resize']:
                return False
            q.epilogue, q.res_in, q.res_in_bits, q.res_out_bits = _lib.EPI_RESIDUAL, 1, 16, 16
        else:
            q.epilogue, q.out_q, q.out_bits, q.relu, q.q_lo, q.q_hi = _lib.EPI_REQUANT, 1, ent['out_bits'], 1, ent['rng'][0], ent['rng'][1]
        if _lib.load().hawq_conv2d_band_tile(C.byref(q)) != 0:
            return True
        # a layer only the round-5 kernels take (planar input, host-packed weight stream): ADVICE r5
        if in_bits != c.w_bits or c.band() is None:
            return False
        q.wgt_band, q.in_planar, q.ctab = c.band().data_ptr(), 1, 1
        if is_last:
            q.res_out_bits, q.flags, q.res_out = 16, 1, 1
        return _lib.load().hawq_conv2d_band2_tile(C.byref(q)) != 0

    def _try_fuse(self, a, u, nxt, N, ho, wo, keep):
        """Expand conv launch `a` of unit `u` (RESIDUAL epilogue, already filled) + reduce conv of unit `nxt`:
        returns (ExpandReduceArgs, output tensor, out_bits, planar) or None if the library does not take the pair."""
        if nxt is None or nxt['resize'] or len(nxt['convs']) != 3 or not a.fast_tables:
            return None
        if int(u['name'].split('.')[0][len('stage'):]) not in self.fuse_stages:
            return None
        ent = nxt['convs'][0]
        c = ent['conv']
        if not ent.get('fast', False) or ent['out_bits'] not in (4, 8) or a.in_bits != 8 or a.w_bits != 8:
            return None
        ob = ent['out_bits']   # 4: the reduce conv's output feeds a nibble 3x3 conv - the fused kernels pack it (hawq4) themselves
        er = _lib.ExpandReduceArgs()
        C.memmove(C.byref(er.expand), C.byref(a), C.sizeof(a))
        er.expand.out_q, er.expand.out_bits = None, 8   # the block input of the next unit stays on chip (its storage width is moot)
        r = er.reduce
        if c.w_bits != 8 and not hasattr(c, "w8"):
            # a 4-bit reduce conv whose input is stored as nibbles when it runs as its own launch: the fused launch reads
            # the same integers from an int8 copy of its weights (the block input never takes a storage format at all)
            c.w8 = torch.from_numpy(packing.pack_conv_weight(c.w_host, 8)).to(self.dev)
        r.wgt, r.bias = (c.w if c.w_bits == 8 else c.w8).data_ptr(), c.bias.data_ptr()
        if c.w_bits == 8 and c.k128() is not None:
            r.wgt_k128 = c.k128().data_ptr()   # (for the two-launch form of the pair; ignored by the fused launch)
        r.N, r.H, r.W, r.Cin, r.Cout, r.KH, r.KW, r.stride, r.pad = N, ho, wo, c.cin, c.cout, c.kh, c.kw, c.stride, c.pad
        r.in_bits, r.w_bits = 8, 8
        r.m, r.e, r.ctab = ent['m'].data_ptr(), ent['e'].data_ptr(), ent['ctab'].data_ptr()
        r.flags = self.flags.data_ptr()
        r.epilogue, r.relu = _lib.EPI_REQUANT, 1
        r.out_bits, r.q_lo, r.q_hi = ob, ent['rng'][0], ent['rng'][1]
        r.fast_tables = (5 if ent.get('tie', False) else 1) | (8 if ent.get('ck0', False) else 0)
        r.out_q = 1  # placeholder for the applicability query
        if _lib.load().hawq_conv_expand_reduce_variants(C.byref(er)) == 0:
            return None
        out = self._alloc(N * ho * wo * c.cout * ob // 8, torch.uint8)
        r.out_q = out.data_ptr()
        planar = self.planar and self._band_takes(nxt['convs'][1], N, ho, wo, ob, False, nxt)
        r.out_planar = int(planar)
        # the same two layers as separate launches (the block input q then goes through memory)
        q = self._alloc(N * ho * wo * c.cin * nxt['a_bits'] // 8, torch.uint8)
        a.out_q = q.data_ptr()
        r1 = _lib.ConvArgs()
        C.memmove(C.byref(r1), C.byref(r), C.sizeof(r1))
        r1.in_, r1.in_bits = q.data_ptr(), nxt['a_bits']
        r1.wgt, r1.w_bits = c.w.data_ptr(), c.w_bits
        # the two-launch form reads the block input at its storage width: the 128-byte K chunks of the weights must be of that width too
        r1.wgt_k128 = c.k128().data_ptr() if (c.k128() is not None and c.w_bits == nxt['a_bits']) else None
        pair = _FusedPair(er, a, r1, self.stream.cuda_stream)
        keep += [out, er, q, r1, pair]
        return pair, out, ob, planar

    def _try_solo(self, a, u, keep):
        """Expand conv launch `a` (RESIDUAL epilogue, single branch, completely filled) as a candidate for the
        wave-private kernel (hawq_conv_expand_reduce with reduce.wgt == NULL): a _FusedPair without a reduce conv, or None."""
        if not a.fast_tables or a.in2 or os.environ.get("HAWQ_NO_SOLO"):
            return None
        er = _lib.ExpandReduceArgs()
        C.memmove(C.byref(er.expand), C.byref(a), C.sizeof(a))
        if _lib.load().hawq_conv_expand_reduce_variants(C.byref(er)) == 0:
            return None
        pair = _FusedPair(er, a, None, self.stream.cuda_stream)
        keep += [er, pair]
        return pair

    def _alloc(self, n, dtype):
        return torch.empty(n, dtype=dtype, device=self.dev)

    def _build(self, N, H, W, x_view=None, logits_view=None):
        """Allocate activation buffers for batch N and record the launch list (choosing the chain count first
        when it was left open)."""
        self._build_plan(N, H, W, x_view, logits_view)
        if x_view is None and not os.environ.get("HAWQ_KEEP_PACKED"):
            self._release_unused_weights()

    def _release_unused_weights(self):
        """Drop the round-5 weight streams (wgt_band / wgt_k128 / wgt2_k128: a second copy of a layer's weights in another order) of
        every layer whose launches - in all chains of the plan just built - run a kernel that does not read them, and clear the
        pointers in those launches' argument blocks (a later build of another batch shape packs them again on demand).
        HAWQ_KEEP_PACKED=1 keeps them (tools/tile_sweep.py switches tiles after the build)."""
        lib = _lib.load()
        n, nb2, ng2 = lib.hawq_conv2d_num_tiles(), lib.hawq_conv2d_num_band2_tiles(), lib.hawq_conv2d_num_gemm2_tiles()
        band_ids, gemm_ids = range(n - ng2 - nb2 + 1, n - ng2 + 1), range(n - ng2 + 1, n + 1)
        args = []
        for e in (self.subs or [self]):
            args += list(getattr(e, "_conv_args", []))
            for pr in getattr(e, "_er_args", []):
                if not pr.fused:
                    args += [pr.expand] + ([pr.reduce] if pr.reduce is not None else [])
        live_band = {a.wgt_band for a in args if a.wgt_band and (a.tile in band_ids or a.tile == 0)}   # (tile 0: the library's own default may pick one)
        live_k128 = {ptr for a in args if (a.tile in gemm_ids or a.tile == 0) for ptr in (a.wgt_k128, a.wgt2_k128) if ptr}
        every = []
        for e in (self.subs or [self]):
            every += list(getattr(e, "_conv_args", []))
            for pr in getattr(e, "_er_args", []):
                every += [pr.expand, pr.er.expand, pr.er.reduce] + ([pr.reduce] if pr.reduce is not None else [])
        for a in every:
            if a.wgt_band and a.wgt_band not in live_band:
                a.wgt_band = None
            if a.wgt_k128 and a.wgt_k128 not in live_k128:
                a.wgt_k128 = None
            if a.wgt2_k128 and a.wgt2_k128 not in live_k128:
                a.wgt2_k128 = None
        convs = [c for u in self.P['units'] for c in [ent['conv'] for ent in u['convs']] + ([u['ident']] if u['resize'] else [])]
        self.packed_weight_bytes = 0
        for c in convs:
            wb, wk = c._w_band, c._w_k128
            c.release(wb is not None and wb.data_ptr() not in live_band, wk is not None and wk.data_ptr() not in live_k128)
            self.packed_weight_bytes += sum(t.numel() for t in (c._w_band, c._w_k128) if t is not None)

    def _build_plan(self, N, H, W, x_view=None, logits_view=None):
        if x_view is None:
            self._plan_on = bool(self.plan) and int(self.plan.get("batch", -1)) == N and not self.keep_acc
            if not self._plan_on and hasattr(self, "chains_req"):
                # a batch shape the plan was not recorded for: the chain count a replay of ANOTHER shape left behind does not carry over
                self.chains = max(1, self.chains_req)
        if x_view is None and self._plan_on:
            try:
                if int(self.plan.get("num_conv_tiles", -1)) != int(_lib.load().hawq_conv2d_num_tiles()):
                    raise StalePlan("recorded for a library with another tile inventory")
                self.chains = max(1, int(self.plan["chains"]))
                self._build_chains(N, H, W, x_view, logits_view)
                names = self.plan.get("conv_launches")
                if names is not None and [n for n in names if n not in self._UNTUNED_SINCE] != list(self.tile_choice.keys()):
                    raise StalePlan("recorded for another launch list (other network, schedule or storage rule)")
                self.plan_source = self.plan.get("source", "replayed a recorded plan")
                return
            except StalePlan as exc:
                print(f"[hawq_amd] recorded plan ignored ({exc}); tuning instead", file=sys.stderr)
                self._plan_on = False
                self.chains = max(1, self.chains_req)
                self.tile_choice, self.er_choice = {}, {}
        if x_view is None and getattr(self, "chains_req", 1) == 0 and self.use_graph and self.autotune:
            timing, plans = {}, {}
            # small batches (what one GPU sees when 128 images are sharded over 4 / 8 ranks) launch fewer workgroups
            # than the chip has CUs in most layers: two concurrent half-batches can still pay, three never did
            for c in ((1, 2, 3) if N >= 48 else ((1, 2) if N >= 8 else (1,))):
                self.chains = c
                self._build_chains(N, H, W)
                timing[c] = self._time_graph() if N >= 8 else 0.0
                plans[c] = self._plan_snapshot()
                self._drop_graph()
            self.chain_timing_ms = timing
            # One tuning run per chain count decided this so far, and one noisy run (timing noise makes two tuning runs differ in a few
            # layers, round 5 saw 1.51 ms for a topology whose plans replay at 1.40 ms) could discard the better topology for good: every
            # chain count within 5 % of the best gets the full set of trials, the fastest replay over all of them wins.
            best = min(timing.values())
            # (at most the two fastest: each candidate costs HAWQ_TUNE_TRIALS full tuning passes - ~20 s each at batch 128 - and the
            #  engines / graphs of discarded candidates are dropped as soon as their plan snapshot is taken)
            order = sorted((c for c in timing if timing[c] <= 1.05 * best), key=timing.get)[:2] if N >= 8 else [min(timing, key=timing.get)]
            winner = None
            for c in order:
                self.chains = c
                self._build_chains(N, H, W, x_view, logits_view)
                if N < 8 or plans.get(c) is None:
                    winner = (0.0, c, None, ())
                    break
                # the rebuild tuned again: keep whichever plan for this chain count replays fastest
                cands = [plans[c], self._plan_snapshot()]
                for _ in range(max(0, int(os.environ.get("HAWQ_TUNE_TRIALS", "4")) - 2)):
                    self._build_chains(N, H, W, x_view, logits_view)
                    cands.append(self._plan_snapshot())
                # all on the final buffers, in two interleaved rounds (a plan's replay time drifts by ~1 % with the clocks: the
                # second round keeps one lucky measurement from deciding), the faster of a plan's two times counts
                times = [float("inf")] * len(cands)
                for _ in range(2):
                    for i, pl in enumerate(cands):
                        self._plan_apply(pl)
                        times[i] = min(times[i], self._time_graph(24))
                if winner is None or min(times) < winner[0]:
                    winner = (min(times), c, cands[times.index(min(times))], tuple(round(t, 4) for t in times))
            if self.chains != winner[1]:
                self.chains = winner[1]
                self._build_chains(N, H, W, x_view, logits_view)
            if winner[2] is not None:
                self._plan_apply(winner[2])
                self.plan_trials_ms = winner[3]
            return
        self._build_chains(N, H, W, x_view, logits_view)

    def _fixed(self, key):
        """Recorded choice for `key` ("chains" / "tiles" / "fused_variants" / "fused_split_tiles"): the plan handed to the constructor
        while the batch shape it was recorded for is being built (`_plan_on`; the chains of a multi-chain engine inherit it), else
        the measurement switches HAWQ_CHAINS / HAWQ_TILES / HAWQ_ER_TILES / HAWQ_ER_SPLIT_TILES, else None (tune)."""
        pl = getattr(self, "plan", None)
        if pl and getattr(self, "_plan_on", False):
            # chains whose tuned choices differ (uneven sub-batches, layers only one tile takes) are recorded one by one
            per = pl.get("per_chain")
            i = getattr(self, "_chain_index", None)
            if per and i is not None and key != "chains":
                if i >= len(per):
                    raise StalePlan(f"the recorded plan lists {len(per)} chains")
                if per[i].get(key) not in (None, ""):
                    return str(per[i][key])
            if pl.get(key) not in (None, ""):
                return str(pl[key])
        return os.environ.get({"chains": "HAWQ_CHAINS", "tiles": "HAWQ_TILES", "fused_variants": "HAWQ_ER_TILES",
                               "fused_split_tiles": "HAWQ_ER_SPLIT_TILES"}[key])

    def export_plan(self):
        """The plan of the batch shape built last, as plain strings (what bench.py prints as config.autotuned_tiles / fused_variants /
        fused_split_tiles / concurrent_sub_batches): feed it back through ``IntegerEngine(model, plan=...)`` to replay it."""
        if self._batch is None:
            raise RuntimeError("export_plan: no batch shape has been built yet")

        def strings(e):
            return {"tiles": ".".join(str(t) for t in e.tile_choice.values()),
                    "fused_variants": ".".join(str(t) for t in e.er_choice.values()),
                    "fused_split_tiles": ".".join(f"{a}.{b}" for a, b in getattr(e, "er_split_tiles", {}).values())}
        per = [strings(e) for e in self.subs]
        # the top-level strings are chain 0's; "per_chain" is only written when another chain runs something else
        extra = {"per_chain": per} if any(p != per[0] for p in per[1:]) else {}
        return {"batch": int(self._batch[0]), "chains": int(self.chains), "expand_in8": self.expand_in8, **extra,
                "tiles": ".".join(str(t) for t in self.tile_choice.values()),
                "fused_variants": ".".join(str(t) for t in self.er_choice.values()),
                "fused_split_tiles": ".".join(f"{a}.{b}" for a, b in getattr(self, "er_split_tiles", {}).values()),
                "conv_launches": list(self.tile_choice.keys()), "pair_launches": list(self.er_choice.keys()),
                # guards against replaying a plan on a build whose kernels are numbered differently
                "num_conv_tiles": int(_lib.load().hawq_conv2d_num_tiles()),
                "pair_variant_counts": [int(_lib.load().hawq_conv_expand_reduce_variants(C.byref(p.er))) for p in (self.subs[0] if self.subs else self)._er_args]}

    def _plan_snapshot(self):
        """Tile / fused-variant choice of every launch of the current plan (one entry per chain)."""
        return [dict(tiles=[a.tile for a in e._conv_args],
                     pairs=[(p.fused, p.er.tile, p.expand.tile, p.reduce.tile if p.reduce is not None else 0) for p in e._er_args],
                     tile_choice=dict(e.tile_choice), er_choice=dict(e.er_choice), er_split=dict(getattr(e, "er_split_tiles", {})))
                for e in (self.subs or [self])]

    def _plan_apply(self, plan):
        engines = self.subs or [self]
        if len(plan) != len(engines) or any(len(pl["tiles"]) != len(e._conv_args) or len(pl["pairs"]) != len(e._er_args)
                                            for pl, e in zip(plan, engines)):
            return
        for pl, e in zip(plan, engines):
            for a, t in zip(e._conv_args, pl["tiles"]):
                a.tile = t
            for p, (fused, vt, te, tr) in zip(e._er_args, pl["pairs"]):
                p.fused, p.er.tile, p.expand.tile = fused, vt, te
                if p.reduce is not None:
                    p.reduce.tile = tr
            e.tile_choice.clear(), e.tile_choice.update(pl["tile_choice"])
            e.er_choice.clear(), e.er_choice.update(pl["er_choice"])
            if hasattr(e, "er_split_tiles"):
                e.er_split_tiles.clear(), e.er_split_tiles.update(pl["er_split"])
        self._drop_graph()

    def _drop_graph(self):
        for attr in ("_graph", "_graph_u8"):
            if getattr(self, attr, None) is not None:
                _lib.call("hawq_graph_destroy", getattr(self, attr))
                setattr(self, attr, None)
        self.x_u8 = None
        self._lut_key = None  # the look-up table lives in buffers that a rebuild replaces

    def _time_graph(self, reps: int = 8) -> float:
        """ms per replay of the captured graph (tuning only) on synthetic images ~ N(0, 1): an all-zero batch switches fewer
        bits in every pipe than real data does and replays ~1.5 % faster, which is not the regime the plans are chosen for."""
        if os.environ.get("HAWQ_TUNE_INPUT", "normal") == "zero":
            self.x_in.zero_()
        else:
            g = torch.Generator(device=self.dev)
            g.manual_seed(0)
            self.x_in.normal_(generator=g)
        e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
        _lib.call("hawq_event_create", C.byref(e0))
        _lib.call("hawq_event_create", C.byref(e1))
        with torch.cuda.stream(self.stream):
            for _ in range(2):
                self.run_resident()
            _lib.call("hawq_event_record", e0, self.stream.cuda_stream)
            for _ in range(reps):
                self.run_resident()
            _lib.call("hawq_event_record", e1, self.stream.cuda_stream)
        torch.cuda.synchronize(self.dev)
        _lib.call("hawq_event_elapsed_ms", e0, e1, C.byref(ms))
        _lib.call("hawq_event_destroy", e0)
        _lib.call("hawq_event_destroy", e1)
        self.flags.zero_()
        return ms.value / reps

    def _build_chains(self, N, H, W, x_view=None, logits_view=None):
        P, dev = self.P, self.dev
        self._drop_graph()
        if self.chains > 1 and N >= 2 * self.chains:
            self.x_in = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
            self.logits = torch.empty(N, P['fc']['nout'], dtype=torch.float32, device=dev)
            self.subs, b0 = [], 0
            # HAWQ_SPLIT="72,56": measurement switch - uneven sub-batches (profiles/r03_uneven_split.md)
            split = [int(v) for v in os.environ.get("HAWQ_SPLIT", "").split(",") if v]
            if len(split) != self.chains or sum(split) != N or min(split) < 1:
                split = [N // self.chains + (1 if i < N % self.chains else 0) for i in range(self.chains)]
            for i in range(self.chains):
                b1 = b0 + split[i]
                sub = IntegerEngine(None, _parent=self)
                sub._chain_index = i
                sub._plan_on = getattr(self, "_plan_on", False)
                sub._build(b1 - b0, H, W, self.x_in[b0:b1], self.logits[b0:b1])
                self.subs.append(sub)
                b0 = b1
            self._ops, self._keep, self._batch, self._graph = _OpList(), [], (N, H, W), None
            if (self.autotune and not self._fixed("tiles") and os.environ.get("HAWQ_JOINT_TUNE", "1") != "0"
                    and all(hasattr(sub, "_tile_times") for sub in self.subs)):
                self._autotune_joint()
            self.n_fast, self.n_conv, self.n_k0, self.n_tie = (self.subs[0].n_fast, self.subs[0].n_conv, self.subs[0].n_k0,
                                                              self.subs[0].n_tie)
            self.n_ck0 = self.subs[0].n_ck0
            self.tile_choice, self.er_choice = self.subs[0].tile_choice, self.subs[0].er_choice
            self.er_split_tiles = getattr(self.subs[0], "er_split_tiles", {})
            return
        self.subs = []
        ops, keep = _OpList(), []
        self._conv_args, self._conv_names = [], []
        self.acc_taps = {}
        self.res_taps = {}   # unit name -> (stored post-ReLU residual tensor of this plan, NHWC shape): parity tests read them back
        self.n_fast = self.n_conv = self.n_k0 = self.n_tie = self.n_ck0 = 0  # how many conv launches run the fast-contract kernels (/ shift-free)
        sp = self.stream.cuda_stream
        ptr = lambda t: None if t is None else t.data_ptr()
        rdt = torch.uint16 if self.res_bits == 16 else torch.int32
        self.x_in = x_view if x_view is not None else torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
        # stem
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        Hp, Wp = 2 * (Ho - 1) + 7 + 1, 2 * (Wo - 1) + 8
        Hp, Wp = max(Hp, H + 3), max(Wp + (Wp & 1), W + 3 + ((W + 3) & 1))
        unfused = self.keep_acc or bool(os.environ.get("HAWQ_UNFUSED_STEM"))
        xq = torch.zeros(N * Hp * Wp * 4, dtype=torch.int8, device=dev) if unfused else None
        st = P['stem']
        stem16 = self._alloc(N * Ho * Wo * 64, torch.uint16) if unfused else None
        stem_acc = self._alloc(N * Ho * Wo * 64, torch.int32) if self.keep_acc else None
        if stem_acc is not None:
            self.acc_taps['stem'] = (stem_acc, (N, Ho, Wo, 64))
        H1, W1 = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        units = P['units']
        u0 = units[0]
        res = self._alloc(N * H1 * W1 * 64, torch.uint16) if not u0['resize'] else None
        qa = self._alloc(N * H1 * W1 * 64 * u0['a_bits'] // 8, torch.uint8)
        c = st['conv']
        if self.keep_acc or os.environ.get("HAWQ_UNFUSED_STEM"):
            # three launches with the 112x112 intermediate in memory (exposes the stem accumulators)
            ops.append(partial(_lib.call, "hawq_quantize_input", self.x_in.data_ptr(), xq.data_ptr(), N, 3, H, W, Hp, Wp,
                               3, 3, P['inv_s_in'], -128, 127, sp))
            ops.append(partial(_lib.call, "hawq_stem_conv7", xq.data_ptr(), c.w.data_ptr(), c.bias.data_ptr(),
                               st['m'].data_ptr(), st['e'].data_ptr(), N, Hp, Wp, Ho, Wo, st['rng'][0], st['rng'][1],
                               stem16.data_ptr(), ptr(stem_acc), sp))
            ops.append(partial(_lib.call, "hawq_maxpool3s2_requant", stem16.data_ptr(), N, Ho, Wo, 64, ptr(res),
                               qa.data_ptr(), u0['a_bits'], u0['mq'], u0['eq'], u0['a_rng'][0], u0['a_rng'][1], sp))
        else:
            ops.append(partial(_lib.call, "hawq_stem_fused", self.x_in.data_ptr(), N, 3, H, W, P['inv_s_in'], -128, 127,
                               c.w.data_ptr(), c.bias.data_ptr(), st['m'].data_ptr(), st['e'].data_ptr(), st['rng'][0],
                               st['rng'][1], ptr(res), qa.data_ptr(), u0['a_bits'], u0['mq'], u0['eq'], u0['a_rng'][0],
                               u0['a_rng'][1], int(st['fast'] and self.fast), sp))
            # everything after (x, lut, N, C, H, W) of the uint8-input twin of this launch (forward_uint8)
            self._stem_u8_tail = (c.w.data_ptr(), c.bias.data_ptr(), st['m'].data_ptr(), st['e'].data_ptr(), st['rng'][0],
                                  st['rng'][1], ptr(res), qa.data_ptr(), u0['a_bits'], u0['mq'], u0['eq'], u0['a_rng'][0],
                                  u0['a_rng'][1], int(st['fast'] and self.fast), sp)
        keep += [xq, stem16, stem_acc, res, qa]
        h, w = H1, W1
        res_bits_in = 16
        fused_in = None
        self._er_args, self._er_names = [], []
        for ui, u in enumerate(units):
            nxt = units[ui + 1] if ui + 1 < len(units) else None
            x_in, x_bits, hin, win, x_planar = qa, u['a_bits'], h, w, False
            for ci, ent in enumerate(u['convs']):
                c = ent['conv']
                if ci == 0 and fused_in is not None:   # this unit's reduce conv ran inside the previous unit's launch
                    x_in, x_bits, x_planar = fused_in
                    hin, win = (hin + 2 * c.pad - c.kh) // c.stride + 1, (win + 2 * c.pad - c.kw) // c.stride + 1
                    fused_in = None
                    self.n_fast += 1
                    self.n_conv += 1
                    continue
                ho, wo = (hin + 2 * c.pad - c.kh) // c.stride + 1, (win + 2 * c.pad - c.kw) // c.stride + 1
                a = _lib.ConvArgs()
                a.in_, a.wgt, a.bias = x_in.data_ptr(), c.w.data_ptr(), c.bias.data_ptr()
                if x_bits == c.w_bits and c.band() is not None:
                    a.wgt_band = c.band().data_ptr()
                if x_bits == c.w_bits and c.k128() is not None:
                    a.wgt_k128 = c.k128().data_ptr()
                a.N, a.H, a.W, a.Cin, a.Cout = N, hin, win, c.cin, c.cout
                a.KH, a.KW, a.stride, a.pad = c.kh, c.kw, c.stride, c.pad
                a.in_bits, a.w_bits = x_bits, c.w_bits
                a.m, a.e = ent['m'].data_ptr(), ent['e'].data_ptr()
                a.flags = self.flags.data_ptr()
                a.fast_tables = int(bool(ent.get('fast', False)) and self.res_bits == 16 and self.fast)
                if a.fast_tables and ent.get('tie', False):
                    a.fast_tables = 5  # some table is not provably tie-free: exact tie handling in the epilogue
                elif a.fast_tables and ent.get('k0', False):
                    a.fast_tables = 3  # no pre-shift anywhere: the shorter requant
                self.n_tie += int(a.fast_tables == 5)
                self.n_k0 += int(a.fast_tables == 3)
                if a.fast_tables and ent.get('ck0', False):
                    a.fast_tables |= 8  # every per-channel pre-shift of this launch's ctab (and ctab_id) is zero
                    self.n_ck0 += 1
                if a.fast_tables:
                    a.ctab = ent['ctab'].data_ptr()
                    if ci == len(u['convs']) - 1 and u['resize']:
                        a.ctab_id = u['ctab_id'].data_ptr()
                self.n_fast += int(a.fast_tables != 0)
                self.n_conv += 1
                a.tile = int(os.environ.get("HAWQ_TILE_RES" if ci == len(u['convs']) - 1 else "HAWQ_TILE_REQ", "0"))
                a.in_planar = int(x_planar)
                tap_name = f"{u['name']}.quant_convbn{ci + 1}"
                if ci < len(u['convs']) - 1:
                    out = self._alloc(N * ho * wo * c.cout * ent['out_bits'] // 8, torch.uint8)
                    a.epilogue, a.relu = _lib.EPI_REQUANT, 1
                    a.out_q, a.out_bits, a.q_lo, a.q_hi = out.data_ptr(), ent['out_bits'], ent['rng'][0], ent['rng'][1]
                    keep.append(out)
                    x_next, xb_next = out, ent['out_bits']
                    # the only reader of `out` is the next conv of this unit: if that is a 3x3 layer a band kernel
                    # takes, write channel-group planes (what the band kernel's LDS-DMA fill reads at full rate)
                    planar_next = bool(a.fast_tables) and self.planar and self._band_takes(u['convs'][ci + 1], N, ho, wo, xb_next,
                                                                                            ci + 1 == len(u['convs']) - 1, u)
                    a.out_planar = int(planar_next)
                else:
                    a.epilogue = _lib.EPI_RESIDUAL
                    if u['resize']:
                        ic = u['ident']
                        a.in2, a.wgt2, a.bias2 = qa.data_ptr(), ic.w.data_ptr(), ic.bias.data_ptr()
                        if u['a_bits'] == ic.w_bits and ic.k128() is not None:
                            a.wgt2_k128 = ic.k128().data_ptr()
                        a.H2, a.W2, a.Cin2, a.stride2 = h, w, ic.cin, ic.stride
                        a.in2_bits, a.w2_bits = u['a_bits'], ic.w_bits
                        a.m_id, a.e_id = u['m_id'].data_ptr(), u['e_id'].data_ptr()
                    else:
                        a.res_in, a.res_in_bits = res.data_ptr(), res_bits_in
                        a.m_id_scalar, a.e_id_scalar = u['m_id_s'], u['e_id_s']
                    need_res = (nxt is None) or (not nxt['resize'])
                    new_res = self._alloc(N * ho * wo * c.cout, rdt) if need_res else None
                    if new_res is not None:
                        a.res_out, a.res_out_bits = new_res.data_ptr(), self.res_bits
                        self.res_taps[u['name']] = (new_res, (N, ho, wo, c.cout))
                    new_qa = None
                    if nxt is not None:
                        a.out_bits = nxt['a_bits']
                        a.q_lo, a.q_hi, a.mq, a.eq = nxt['a_rng'][0], nxt['a_rng'][1], nxt['mq'], nxt['eq']
                    keep.append(new_res)
                if self.keep_acc:
                    self._add_acc_tap(ops, keep, a, tap_name, N, ho, wo, c.cout)
                    if ci == len(u['convs']) - 1 and u['resize']:
                        self._add_ident_tap(ops, keep, u, qa, N, h, w, ho, wo)
                keep.append(a)
                fz = self._try_fuse(a, u, nxt, N, ho, wo, keep) if (ci == len(u['convs']) - 1 and not self.keep_acc) else None
                if fz is None and ci == len(u['convs']) - 1 and nxt is not None:
                    new_qa = self._alloc(N * ho * wo * c.cout * nxt['a_bits'] // 8, torch.uint8)
                    a.out_q = new_qa.data_ptr()
                    keep.append(new_qa)
                if fz is not None:
                    pair, fout, fbits, fplanar = fz
                    fused_in = (fout, fbits, fplanar)
                    self._er_args.append(pair)
                    self._er_names.append(tap_name)
                    ops.next_name = tap_name + "+" + nxt['name'] + ".quant_convbn1"
                    ops.append(pair)
                    continue
                solo = self._try_solo(a, u, keep) if (ci == len(u['convs']) - 1 and not self.keep_acc) else None
                if solo is not None:
                    self._er_args.append(solo)
                    self._er_names.append(tap_name + "@solo")
                    ops.next_name = tap_name
                    ops.append(solo)
                    continue
                self._conv_args.append(a)
                self._conv_names.append(tap_name)
                ops.next_name = tap_name + ("+identity" if (a.in2 is not None) else "")
                ops.append(partial(_lib.call, "hawq_conv2d", C.byref(a), sp))
                if ci < len(u['convs']) - 1:
                    x_in, x_bits, hin, win, x_planar = x_next, xb_next, ho, wo, planar_next
            res, qa, h, w = new_res, new_qa, ho, wo
            res_bits_in = self.res_bits
        cl = units[-1]['convs'][-1]['conv'].cout
        qf = self._alloc(N * cl, torch.int8)
        pooled = self._alloc(N * cl, torch.int32) if self.keep_acc else None
        if pooled is not None:
            self.acc_taps['final_pool'] = (pooled, (N, cl))
        o = P['out']
        ops.append(partial(_lib.call, "hawq_avgpool_requant", res.data_ptr(), self.res_bits, N, h * w, cl, qf.data_ptr(),
                           ptr(pooled), o['mq'], o['eq'], o['rng'][0], o['rng'][1], sp))
        fc = P['fc']
        if fc['k'] != cl:
            raise RuntimeError("FC input width does not match the last stage")
        self.logits = logits_view if logits_view is not None else torch.empty(N, fc['nout'], dtype=torch.float32, device=dev)
        a = _lib.ConvArgs()
        a.in_, a.wgt, a.bias = qf.data_ptr(), fc['w'].data_ptr(), fc['bias'].data_ptr()
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = N, 1, 1, fc['k'], fc['nout_p'], 1, 1, 1, 0
        a.in_bits, a.w_bits = 8, 8
        a.epilogue = _lib.EPI_DEQUANT
        a.out_f32, a.fscale, a.ldo, a.n_valid = self.logits.data_ptr(), fc['fscale'].data_ptr(), fc['nout'], fc['nout']
        if self.keep_acc:
            self._add_acc_tap(ops, keep, a, 'quant_output', N, 1, 1, fc['nout_p'])
        ops.next_name = "quant_output"
        if _lib.load().hawq_fc_dequant_ok(N, fc['k'], fc['nout_p']) and not os.environ.get("HAWQ_NO_FC2"):
            # round 5: the classifier's own kernel (fc_dequant.hip; K split over the waves of a workgroup, no LDS staging) - the same bytes as the
            # DEQUANT epilogue of hawq_conv2d, nothing to tune
            ops.append(partial(_lib.call, "hawq_fc_dequant", qf.data_ptr(), fc['w'].data_ptr(), fc['bias'].data_ptr(), fc['fscale'].data_ptr(),
                               self.logits.data_ptr(), N, fc['k'], fc['nout_p'], fc['nout'], fc['nout'], sp))
        else:
            ops.append(partial(_lib.call, "hawq_conv2d", C.byref(a), sp))
            self._conv_args.append(a)  # the FC GEMM (M = batch, K = 2048) is tile-tuned like the convs
            self._conv_names.append("quant_output")
        keep += [qf, pooled, a]
        self._ops, self._keep, self._batch = ops, keep, (N, H, W)
        self._graph = None
        if self.autotune and not self.keep_acc:
            self._autotune_tiles()

    def _add_acc_tap(self, ops, keep, a, name, N, ho, wo, cout):
        """Extra RAW launch of the same conv to expose its int32 accumulators (tests only)."""
        acc = self._alloc(N * ho * wo * cout, torch.int32)
        r = _lib.ConvArgs()
        C.memmove(C.byref(r), C.byref(a), C.sizeof(r))
        r.in2 = None
        r.epilogue, r.out_acc = _lib.EPI_RAW, acc.data_ptr()
        keep += [acc, r]
        self.acc_taps[name] = (acc, (N, ho, wo, cout))
        ops.append(partial(_lib.call, "hawq_conv2d", C.byref(r), self.stream.cuda_stream))

    def _add_ident_tap(self, ops, keep, u, qa, N, h, w, ho, wo):
        ic = u['ident']
        acc = self._alloc(N * ho * wo * ic.cout, torch.int32)
        r = _lib.ConvArgs()
        r.in_, r.wgt, r.bias = qa.data_ptr(), ic.w.data_ptr(), ic.bias.data_ptr()
        r.N, r.H, r.W, r.Cin, r.Cout, r.KH, r.KW, r.stride, r.pad = N, h, w, ic.cin, ic.cout, 1, 1, ic.stride, 0
        r.in_bits, r.w_bits = u['a_bits'], ic.w_bits
        r.epilogue, r.out_acc = _lib.EPI_RAW, acc.data_ptr()
        keep += [acc, r]
        self.acc_taps[u['name'] + ".quant_identity_convbn"] = (acc, (N, ho, wo, ic.cout))
        ops.append(partial(_lib.call, "hawq_conv2d", C.byref(r), self.stream.cuda_stream))

    def _autotune_tiles(self, reps: int = 3):
        """Time every conv launch with each tile configuration (HIP events on the engine stream, the
        real buffers, results are identical for every tile) and keep the fastest.  Runs once per batch
        shape, before the hipGraph is captured: ~50 launches x 4 tiles x reps, a few milliseconds."""
        n_tiles = _lib.load().hawq_conv2d_num_tiles()
        sp = self.stream.cuda_stream
        self._tile_times, self._er_times = {}, {}
        # tile ids the tuner no longer tries: never chosen in any plan recorded in rounds 4-6 nor in this round's
        # fresh tuning runs (profiles/r06_retired_tiles.md).  The kernels stay in the library under their ids - recorded plans, the
        # parity tests and tools/tile_sweep.py still reach them - but a tuning pass no longer times them.  HAWQ_RETIRED_TILES="" re-enables.
        retired = {int(v) for v in os.environ.get("HAWQ_RETIRED_TILES", "5,19").split(",") if v}
        # replay a recorded choice, no timing: the constructor's plan for this batch size, or HAWQ_TILES (dotted list as bench.py prints it).
        # A chain of a multi-chain engine replays the plan of the WHOLE batch it is a part of
        fixed = self._fixed("tiles")
        if fixed:
            ids = [int(v) for v in fixed.split(".")]
            names = (self.plan or {}).get("conv_launches") if getattr(self, "_plan_on", False) else None
            if names is not None and len(names) == len(ids) and list(names) != list(self._conv_names):
                # a plan recorded with launches this build no longer tunes (_UNTUNED_SINCE: the classifier since round 5): replay by NAME.
                # The ONLY relaxation: the recorded list minus those launches must be this build's list exactly - a superset recorded
                # for a larger network (resnet101's plan on resnet50) is stale, not replayable
                if [n for n in names if n not in self._UNTUNED_SINCE] != list(self._conv_names):
                    raise StalePlan("recorded for another launch list (other network, schedule or storage rule)")
                by_name = dict(zip(names, ids))
                ids = [by_name[n] for n in self._conv_names]
            if len(ids) != len(self._conv_args):
                raise StalePlan(f"the recorded plan lists {len(ids)} tiles, this plan has {len(self._conv_args)} conv launches")
            counts = self.plan.get("pair_variant_counts") if getattr(self, "_plan_on", False) else None
            if counts is not None and list(counts) != [int(_lib.load().hawq_conv_expand_reduce_variants(C.byref(p.er))) for p in self._er_args]:
                raise StalePlan("the fused expand(-> reduce) kernels of this build are numbered differently")
            for name, a, tid in zip(self._conv_names, self._conv_args, ids):
                a.tile = tid
                self.tile_choice[name] = tid
            split = self._fixed("fused_split_tiles")
            split = [int(x) for x in split.split(".")] if split else None
            self.er_split_tiles = getattr(self, "er_split_tiles", {})
            for k, (name, pair) in enumerate(zip(self._er_names, self._er_args)):
                v = (self._fixed("fused_variants") or ".".join(["1"] * len(self._er_args))).split(".")[k]
                pair.er.tile = int(v)
                pair.fused = pair.er.tile != 0
                if split is not None:   # tiles of the pair's two-launch form (what runs when the recorded variant is 0)
                    te, tr = split[2 * k:2 * k + 2]
                    pair.expand.tile = te
                    if pair.reduce is not None:
                        pair.reduce.tile = tr
                    self.er_split_tiles[name] = (te, tr)
                elif not pair.fused:
                    raise StalePlan("the recorded plan runs a pair as two launches but lists no tiles for them")
                self.er_choice[name] = pair.er.tile
            return
        # the events are created only on the timing path (the replay branch above returns or raises StalePlan without them)
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.call("hawq_event_create", C.byref(e0))
        _lib.call("hawq_event_create", C.byref(e1))
        ms = C.c_float()
        with torch.cuda.stream(self.stream):
            self._launch_all()  # every buffer holds valid data
            for name, a in [(n, k) for n, k in zip(self._conv_names, self._conv_args)]:
                if os.environ.get("HAWQ_TILE_RES") or os.environ.get("HAWQ_TILE_REQ"):
                    break
                times = {}
                for rnd in range(2):  # two rounds, per-tile minimum: one hiccup must not decide a layer's tile
                    for tile in range(1, n_tiles + 1):
                        if (rnd and tile not in times) or tile in retired:
                            continue
                        a.tile = tile
                        try:
                            _lib.call("hawq_conv2d", C.byref(a), sp)  # warm
                            _lib.call("hawq_event_record", e0, sp)
                            for _ in range(reps):
                                _lib.call("hawq_conv2d", C.byref(a), sp)
                            _lib.call("hawq_event_record", e1, sp)
                            _lib.call("hawq_event_elapsed_ms", e0, e1, C.byref(ms))
                        except RuntimeError:
                            continue
                        times[tile] = min(times.get(tile, ms.value), ms.value)
                best_t = min(times, key=times.get)
                self._tile_times[name] = dict(times)
                log = [f"{t}:{v / reps * 1e3:.1f}" for t, v in times.items()]
                if os.environ.get("HAWQ_AUTOTUNE_LOG"):
                    print(f"[autotune N={a.N}] {name}: best {best_t}  us per tile: {' '.join(log)}", file=sys.stderr)
                a.tile = best_t
                self.tile_choice[name] = best_t
            def best_tile(a):   # fastest applicable tile of one hawq_conv2d launch: (tile, ms per `reps` launches)
                times = {}
                for rnd in range(2):
                    for tile in range(1, n_tiles + 1):
                        if (rnd and tile not in times) or tile in retired:
                            continue
                        a.tile = tile
                        try:
                            _lib.call("hawq_conv2d", C.byref(a), sp)
                            _lib.call("hawq_event_record", e0, sp)
                            for _ in range(reps):
                                _lib.call("hawq_conv2d", C.byref(a), sp)
                            _lib.call("hawq_event_record", e1, sp)
                            _lib.call("hawq_event_elapsed_ms", e0, e1, C.byref(ms))
                        except RuntimeError:
                            continue
                        times[tile] = min(times.get(tile, ms.value), ms.value)
                t = min(times, key=times.get)
                return t, times[t]

            fixed_er = os.environ.get("HAWQ_ER_TILES")   # dotted list as printed by bench.py; 0 = two separate launches
            for k, (name, pair) in enumerate(zip(self._er_names, self._er_args)):
                er = pair.er
                if fixed_er:
                    er.tile = int(fixed_er.split(".")[k])
                    pair.fused = er.tile != 0
                    if not pair.fused:
                        te, tr = (int(v) for v in os.environ["HAWQ_ER_SPLIT_TILES"].split(".")[2 * k:2 * k + 2])
                        pair.expand.tile = te
                        if pair.reduce is not None:
                            pair.reduce.tile = tr
                    self.er_choice[name] = er.tile
                    continue
                nvar = _lib.load().hawq_conv_expand_reduce_variants(C.byref(er))
                times = {}
                for rnd in range(2):
                    for tile in range(1, nvar + 1):
                        er.tile = tile
                        _lib.call("hawq_conv_expand_reduce", C.byref(er), sp)
                        _lib.call("hawq_event_record", e0, sp)
                        for _ in range(reps):
                            _lib.call("hawq_conv_expand_reduce", C.byref(er), sp)
                        _lib.call("hawq_event_record", e1, sp)
                        _lib.call("hawq_event_elapsed_ms", e0, e1, C.byref(ms))
                        times[tile] = min(times.get(tile, ms.value), ms.value)
                er.tile = min(times, key=times.get)
                self._er_times[name] = dict(times)
                te, ms_e = best_tile(pair.expand)
                tr, ms_r = best_tile(pair.reduce) if pair.reduce is not None else (0, 0.0)
                pair.expand.tile = te
                if pair.reduce is not None:
                    pair.reduce.tile = tr
                pair.fused = times[er.tile] <= ms_e + ms_r
                if os.environ.get("HAWQ_AUTOTUNE_LOG"):
                    log = [f"{t}:{v / reps * 1e3:.1f}" for t, v in times.items()]
                    print(f"[autotune N={er.expand.N}] {name}+next reduce: fused variants (us) {' '.join(log)} | separate "
                          f"{ms_e / reps * 1e3:.1f} (tile {te}) + {ms_r / reps * 1e3:.1f} (tile {tr}) -> {'fused' if pair.fused else 'separate'}",
                          file=sys.stderr)
                self.er_choice[name] = er.tile if pair.fused else 0
                self.er_split_tiles = getattr(self, "er_split_tiles", {})
                self.er_split_tiles[name] = (te, tr)
        _lib.call("hawq_event_destroy", e0)
        _lib.call("hawq_event_destroy", e1)
        torch.cuda.synchronize(self.dev)
        self.flags.zero_()  # tuning launches ran on whatever the buffers held; only real forwards may raise the flag

    def _autotune_joint(self, reps: int = 4, slack: float = 1.6, top: int = 6):
        """Second tuning pass of a plan with concurrent sub-batch chains.  Each chain has picked its tiles by timing its
        launches ALONE on the chip; in the real forward the same layer of the other chain(s) runs beside it, and what
        then counts is how well the two launches share a CU (LDS footprint, waves, issue slots), not the isolated time:
        a 77 KiB-per-workgroup persistent kernel that is 16 % faster alone made the forward 2.6 % slower with two
        workgroups per CU and 0.5 % faster with one (profiles/r02_band_persist.md).  So: per layer, every tile whose
        isolated time is within `slack` of the best is timed again with ALL chains launching that layer at once (each on its own
        stream), and the tile with the shortest joint time wins - the same for a fused pair against its two-launch form."""
        subs = self.subs
        lib_call = _lib.call
        slack, top = float(os.environ.get("HAWQ_JOINT_SLACK", slack)), int(os.environ.get("HAWQ_JOINT_TOP", top))

        def joint_ms(launch):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = None
            for rnd in range(4):   # first round warms up, then the minimum of three
                start.record(self.stream)
                for sub in subs:
                    sub.stream.wait_event(start)
                    for _ in range(reps):
                        launch(sub)
                    j = torch.cuda.Event()
                    j.record(sub.stream)
                    self.stream.wait_event(j)
                end.record(self.stream)
                end.synchronize()
                if rnd:
                    t = start.elapsed_time(end)
                    best = t if best is None else min(best, t)
            return best

        log = bool(os.environ.get("HAWQ_AUTOTUNE_LOG"))
        with torch.cuda.stream(self.stream):
            n_conv = len(subs[0]._conv_args)
            if any(len(sub._conv_args) != n_conv or sub._conv_names != subs[0]._conv_names for sub in subs):
                return
            for k, name in enumerate(subs[0]._conv_names):
                iso = {}
                for sub in subs:   # a tile must be known to every chain; isolated time = the slowest chain's
                    for t, v in sub._tile_times.get(name, {}).items():
                        iso[t] = max(iso.get(t, 0.0), v)
                iso = {t: v for t, v in iso.items() if all(t in sub._tile_times.get(name, {}) for sub in subs)}
                if len(iso) < 2:
                    continue
                lim = slack * min(iso.values())
                cand = sorted((t for t, v in iso.items() if v <= lim), key=iso.get)[:top]
                res = {}
                for t in cand:
                    for sub in subs:
                        sub._conv_args[k].tile = t
                    res[t] = joint_ms(lambda sub: lib_call("hawq_conv2d", C.byref(sub._conv_args[k]), sub.stream.cuda_stream))
                best = min(res, key=res.get)
                if log:
                    print(f"[joint tune] {name}: " + " ".join(f"{t}:{iso[t] / reps * 1e3:.1f}/{res[t] / reps * 1e3:.1f}" for t in cand)
                          + f" (us alone / all chains) -> {best}", file=sys.stderr)
                for sub in subs:
                    sub._conv_args[k].tile = best
                    sub.tile_choice[name] = best
            n_er = len(subs[0]._er_args)
            if all(len(sub._er_args) == n_er and sub._er_names == subs[0]._er_names for sub in subs):
                for k, name in enumerate(subs[0]._er_names):
                    res = {}
                    variants = set.intersection(*(set(sub._er_times.get(name, {})) for sub in subs))
                    for v in sorted(variants):
                        for sub in subs:
                            sub._er_args[k].er.tile, sub._er_args[k].fused = v, True
                        res[v] = joint_ms(lambda sub: sub._er_args[k]())
                    for sub in subs:   # the two-launch form with each chain's own best tiles
                        sub._er_args[k].fused = False
                    res[0] = joint_ms(lambda sub: sub._er_args[k]())
                    best = min(res, key=res.get)
                    if log:
                        print(f"[joint tune] {name}+next reduce: " + " ".join(f"{v}:{t / reps * 1e3:.1f}" for v, t in sorted(res.items()))
                              + f" (us, all chains; 0 = two launches) -> {best}", file=sys.stderr)
                    for sub in subs:
                        pair = sub._er_args[k]
                        pair.fused = best != 0
                        if best:
                            pair.er.tile = best
                        sub.er_choice[name] = best
        torch.cuda.synchronize(self.dev)
        for sub in subs:
            sub.flags.zero_()

    # ------------------------------------------------------------------ execution
    def _launch_all(self, u8: bool = False):
        if self.subs:  # fork: every chain on its own stream, joined back into self.stream
            fork = torch.cuda.Event()
            fork.record(self.stream)
            for sub in self.subs:
                sub.stream.wait_event(fork)
                sub._launch_all(u8)
                join = torch.cuda.Event()
                join.record(sub.stream)
                self.stream.wait_event(join)
            return
        for i, op in enumerate(self._ops):
            if u8 and i == 0:
                self._stem_u8_op()
            else:
                op()

    def __call__(self, x):
        """fp32 NCHW images on the GPU -> a FRESH fp32 logits tensor.  With the uint16 residual plan the sticky
        overflow flag is read back after the forward (one stream synchronisation) and an overflowing batch is
        transparently recomputed with int32 residuals: the reference never clamps there (quant_utils.py:456)."""
        if not x.is_cuda:
            raise RuntimeError("IntegerEngine: input must be on the MI355X (no CPU path)")
        N, Cc, H, W = x.shape
        if Cc != 3:
            raise ValueError("expected [N,3,H,W] images")
        if self._batch != (N, H, W):
            self._build(N, H, W)
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x_in.copy_(x, non_blocking=True)
            self.run_resident()
            out = self._checked_logits(lambda wide: wide(x))
        cur.wait_stream(self.stream)
        return out

    def _checked_logits(self, redo):
        """Clone of ``self.logits`` for the forward just queued on ``self.stream`` - or, if that forward saturated a
        uint16 residual, the logits of ``redo(int32-residual engine)``.  Called inside ``torch.cuda.stream(self.stream)``."""
        out = self.logits.clone()
        if self.res_bits != 16:
            return out
        self._flag_host.copy_(self.flags, non_blocking=True)
        self.stream.synchronize()
        if not (int(self._flag_host.item()) & 1):
            return out
        self.flags.zero_()
        self.overflow_fallbacks += 1
        self.stream.synchronize()
        return redo(self.wide_engine())

    def wide_engine(self):
        """The int32-residual twin of this plan (exact general kernels on the residual path; built on first use)."""
        if self._wide is None:
            self._wide = IntegerEngine(self.model, residual_bits=32, from_buffers=self.from_buffers, use_graph=False,
                                       fast=self.fast, autotune=False, chains=1)
        return self._wide

    def run_resident(self, u8: bool = False):
        """One forward over ``self.x_in`` (or, ``u8``, over ``self.x_u8``) already resident, on ``self.stream``."""
        if self.use_graph:
            attr = "_graph_u8" if u8 else "_graph"
            if getattr(self, attr, None) is None:
                self._launch_all(u8)  # warm-up outside capture (module loading, first-touch)
                torch.cuda.synchronize(self.dev)
                _lib.call("hawq_graph_begin", self.stream.cuda_stream)
                try:
                    self._launch_all(u8)
                finally:
                    g = C.c_void_p()
                    _lib.call("hawq_graph_end", self.stream.cuda_stream, C.byref(g))
                setattr(self, attr, g)
            _lib.call("hawq_graph_launch", getattr(self, attr), self.stream.cuda_stream)
        else:
            self._launch_all(u8)

    # ------------------------------------------------------------------ uint8 image input (quant_train.py:432-440)
    def input_lut(self, mean, std) -> torch.Tensor:
        """int8 [3][256]: lut[c][u] = QuantAct_input(Normalize_c(ToTensor(u))) with the reference pipeline's own float32
        operations on the host (torchvision ToTensor ``u.float().div(255)``, Normalize ``sub(mean).div(std)``,
        then ``clamp(rint(fl(1/S) * v))``, quant_utils.py:73-97) - so the table look-up in the stem kernel is
        bit-identical to quantising the normalised fp32 tensor."""
        from .quant_utils import input_quant_lut
        return input_quant_lut(self.P['inv_s_in'], mean, std)

    def _ensure_u8(self, N, H, W, x_view=None, lut=None):
        if self.subs:
            if getattr(self, "x_u8", None) is None or self.x_u8.shape[0] != N:
                self.x_u8 = torch.empty(N, H, W, 3, dtype=torch.uint8, device=self.dev)
                self.lut_dev = torch.zeros(3, 256, dtype=torch.int8, device=self.dev)
                self._lut_key = None
                b0 = 0
                for sub in self.subs:
                    n = sub._batch[0]
                    sub._ensure_u8(n, H, W, self.x_u8[b0:b0 + n], self.lut_dev)
                    b0 += n
            return
        if x_view is None:
            if getattr(self, "x_u8", None) is not None and self.x_u8.shape[0] == N:
                return
            x_view = torch.empty(N, H, W, 3, dtype=torch.uint8, device=self.dev)
            lut = torch.zeros(3, 256, dtype=torch.int8, device=self.dev)
            self._lut_key = None
        if not hasattr(self, "_stem_u8_tail"):
            raise RuntimeError("uint8 input needs the fused stem (not available with keep_accumulators / HAWQ_UNFUSED_STEM)")
        self.x_u8, self.lut_dev = x_view, lut
        self._stem_u8_op = partial(_lib.call, "hawq_stem_fused_u8", x_view.data_ptr(), lut.data_ptr(), N, 3, H, W,
                                   *self._stem_u8_tail)

    def forward_uint8(self, x_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """uint8 NHWC images [N,H,W,3] (decoder output, after resize / crop) -> fp32 logits.  Equivalent, bit for bit,
        to ``self(normalised fp32 NCHW tensor)`` as the reference's data pipeline would have built it; the fp32 tensor
        (4x the bytes) never exists."""
        if not x_u8.is_cuda or x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or x_u8.shape[3] != 3:
            raise ValueError("expected a uint8 NHWC [N,H,W,3] tensor on the MI355X")
        N, H, W, _ = x_u8.shape
        if self._batch != (N, H, W):
            self._build(N, H, W)
        self._ensure_u8(N, H, W)
        key = (tuple(float(v) for v in mean), tuple(float(v) for v in std))
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if getattr(self, "_lut_key", None) != key:
                self.lut_dev.copy_(self.input_lut(mean, std).to(self.dev), non_blocking=False)
                self._lut_key = key
            self.x_u8.copy_(x_u8, non_blocking=True)
            self.run_resident(u8=True)
            out = self._checked_logits(lambda wide: wide.forward_uint8(x_u8, mean, std))
        cur.wait_stream(self.stream)
        return out

    def profile_ops(self, repeats: int = 5):
        """Per-launch durations (ms, median of ``repeats``) measured with HIP events around each
        eager launch on the engine stream.  Returns [(name, ms)] in launch order."""
        if self.subs:
            return self.subs[0].profile_ops(repeats)
        sp = self.stream.cuda_stream
        evs = []
        for _ in range(len(self._ops) + 1):
            e = C.c_void_p()
            _lib.call("hawq_event_create", C.byref(e))
            evs.append(e)
        samples = [[] for _ in self._ops]
        for _ in range(repeats):
            _lib.call("hawq_event_record", evs[0], sp)
            for i, op in enumerate(self._ops):
                op()
                _lib.call("hawq_event_record", evs[i + 1], sp)
            for i in range(len(self._ops)):
                ms = C.c_float()
                _lib.call("hawq_event_elapsed_ms", evs[i], evs[i + 1], C.byref(ms))
                samples[i].append(ms.value)
        for e in evs:
            _lib.call("hawq_event_destroy", e)
        return [(n, sorted(v)[len(v) // 2]) for n, v in zip(self._ops.names, samples)]

    def overflowed(self) -> bool:
        """True if a uint16 residual saturated in a ``run_resident`` forward that nobody has handled yet.  ``__call__`` /
        ``forward_uint8`` check and clear the flag themselves (``overflow_fallbacks`` counts the batches they redid with
        int32 residuals); callers of the raw ``run_resident`` (bench.py) read it here."""
        return bool(self.flags.item() & 1)

    def residual(self, unit):
        """The stored post-ReLU residual of `unit` ("stage2.unit3": the value of its quant_act_int32 after ReLU, quant_utils.py:456)
        from the LAST forward of this plan, as an NCHW int32 numpy array over the whole batch (sub-batch chains concatenated);
        None if the plan never stores it (the next unit opens a stage and reads only the 8-bit block input)."""
        parts = []
        for e in (self.subs or [self]):
            if unit not in e.res_taps:
                return None
            t, shp = e.res_taps[unit]
            parts.append(t.cpu().numpy().reshape(shp).astype(np.int32).transpose(0, 3, 1, 2))
        return np.concatenate(parts, 0)

    def accumulators(self, name):
        """int32 NHWC accumulators of a tapped conv (keep_accumulators=True) as an NCHW numpy array."""
        acc, shp = self.acc_taps[name]
        a = acc.cpu().numpy().reshape(shp)
        return a.transpose(0, 3, 1, 2) if len(shp) == 4 else a

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:
            pass
