"""ctypes loader for the in-tree C-ABI library (candle_vllm_amd/libmi355vllm.so).

The product path has NO fallback: if the library is missing or a symbol declared in
include/mi355_vllm.h is not exported, importing this module raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LIB_PATH") or os.path.join(_HERE, "libmi355vllm.so")   # env override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mi355_vllm.h")


def declared_symbols(header_path=HEADER_PATH):
    """Every function name declared in the public header."""
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:(?:const )?void\*?|int|int32_t|int64_t|uint64_t|float\*)\s+(\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
        "there is no CPU fallback for the MI355X decode path")

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

_missing = [s for s in declared_symbols() if not hasattr(lib, s)]
if _missing:
    raise ImportError(f"{LIB_PATH} does not export: {_missing}")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class QmmDesc(ctypes.Structure):
    """mirror of `mi355_qmm_desc` (include/mi355_vllm.h)"""
    _fields_ = [
        ("nseg", c_i32), ("w_tiles", c_vp * 3), ("ggml_type", c_i32 * 3), ("n_rows", c_i32 * 3),
        ("x", c_vp), ("x_dtype", c_i32), ("ldx", c_i32), ("k", c_i32), ("num_tokens", c_i32),
        ("norm_weight", c_vp), ("norm_eps", c_f32), ("epilogue", c_i32),
        ("out", c_vp), ("ldo", c_i32), ("residual", c_vp), ("bias", c_vp),
        ("cos_table", c_vp), ("sin_table", c_vp), ("positions", c_vp), ("slot_mapping", c_vp),
        ("q_out", c_vp), ("key_cache", c_vp), ("value_cache", c_vp),
        ("num_heads", c_i32), ("num_kv_heads", c_i32), ("head_dim", c_i32), ("rotary_dim", c_i32),
        ("block_size", c_i32), ("kv_layout", c_i32),
        ("moe_expert_ids", c_vp), ("moe_pairs", c_i32), ("moe_x_div", c_i32), ("moe_expert_stride", c_i64 * 3),
        ("chain_next", c_i32), ("chain_next_k", c_i32), ("chain_next_norm", c_vp),
        ("rows_dev", c_vp), ("rows_min", c_i32),
        ("group_count", c_i32), ("group_x_stride", c_i64), ("group_out_stride", c_i64),
        ("group_block_table", c_vp),
    ]


class RopeScaling(ctypes.Structure):
    """mirror of `mi355_rope_scaling`"""
    _fields_ = [("type", c_i32), ("factor", ctypes.c_double), ("low_freq_factor", ctypes.c_double),
                ("high_freq_factor", ctypes.c_double), ("original_max_position_embeddings", ctypes.c_double),
                ("alpha", ctypes.c_double), ("beta_fast", ctypes.c_double), ("beta_slow", ctypes.c_double),
                ("attn_factor", ctypes.c_double), ("extrapolation_factor", ctypes.c_double)]


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


for _n in ("copy_blocks_bf16", "copy_blocks_f16", "copy_blocks_f32", "copy_blocks_u8"):
    _sig(_n, None, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_swap_blocks", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i64])
_sig("mi355_reshape_and_cache", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 6 + [c_i64])
_sig("mi355_paged_attention_v1", ctypes.c_int,
     [c_vp] * 6 + [c_i32] * 7 + [c_f32, c_f32, c_i32, c_i32, c_i64])
_sig("mi355_paged_attention_v2", ctypes.c_int,
     [c_vp] * 9 + [c_i32] * 8 + [c_f32, c_f32, c_i32, c_i32, c_i64])
_sig("mi355_prefill_attention", ctypes.c_int,
     [c_vp] * 9 + [c_i32] * 7 + [c_f32, c_f32, c_i32, c_i32, c_i64])
_sig("mi355_paged_attention_window", ctypes.c_int,
     [c_vp] * 6 + [c_i32] * 7 + [c_f32, c_f32, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_prefill_attention_window", ctypes.c_int,
     [c_vp] * 9 + [c_i32] * 7 + [c_f32, c_f32, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_reshape_and_cache_fp8", ctypes.c_int, [c_vp] * 5 + [c_i32] * 5 + [c_f32, c_f32, c_i64])
_sig("mi355_paged_attention_fp8", ctypes.c_int, [c_vp] * 9 + [c_i32] * 8 + [c_f32] * 4 + [c_i64])
_sig("mi355_prefill_attention_fp8", ctypes.c_int, [c_vp] * 7 + [c_i32] * 7 + [c_f32] * 4 + [c_i32, c_i64])
_sig("mi355_rope_inplace", ctypes.c_int, [c_vp] * 5 + [c_i32] * 7 + [c_i64])
_sig("mi355_rms_norm", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_i32, c_i64])
_sig("mi355_silu_mul", ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64])
_sig("mi355_add_f32", ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64])
_sig("mi355_cast", ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i64])
_sig("mi355_embedding_f32", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64])
_sig("mi355_argmax_f32", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64])
_sig("mi355_dequantize", ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i64])
_sig("mi355_qmatmul_ref", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_qweight_repacked_size", c_i64, [c_i32, c_i64, c_i64])
_sig("mi355_qweight_repack", ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i64])
_sig("mi355_qmatmul", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64])
_sig("mi355_qmatmul_fused", ctypes.c_int, [ctypes.POINTER(QmmDesc), c_i64])
_sig("mi355_set_tuning", None, [c_i32, c_i32])
_sig("mi355_debug_set_timestamps", ctypes.c_int, [c_vp])
_sig("mi355_qmv_error", ctypes.c_int, [ctypes.POINTER(ctypes.c_int32), c_i32])
_sig("mi355_qmv_chain_sync_bytes", ctypes.c_int, [])
_sig("mi355_qmatmul_chain", ctypes.c_int, [ctypes.POINTER(QmmDesc), c_i32, c_vp, c_i64])
_sig("mi355_moe_route", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_combine", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_gather", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_group", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_gather_pos", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_group_blocks", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_moe_scatter_combine", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
for _n in ("marlin_4bit_f16", "marlin_4bit_bf16", "marlin_awq_4bit_f16", "marlin_awq_4bit_bf16"):
    _sig(_n, None, [c_vp] * 6 + [c_i32] * 3 + [c_vp, c_i32, c_i64])
_sig("gemm_half_q_half_alt", None, [c_vp] * 6 + [c_i32] * 4 + [c_i64])
_sig("gptq_repack", None, [c_vp, c_vp, c_i32, c_i32, c_i64])
_sig("awq_repack", None, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_marlin_scale_pos", c_i32, [c_i32, c_i32])
_sig("mi355_get_tuning", c_i32, [c_i32])
_sig("mi355_last_error", ctypes.c_int, [])
_sig("mi355_clear_error", None, [])
_sig("mi355_marlin_format_repack", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64])
_sig("mi355_marlin_weight_perm", c_i32, [c_i32])
_sig("mi355_marlin_zero_pos", c_i32, [c_i32])
_sig("mi355_linear", ctypes.c_int, [c_vp] * 5 + [c_i32] * 5 + [c_i64])
_sig("mi355_linear_tiled", ctypes.c_int, [c_vp] * 5 + [c_i32] * 5 + [c_i64])
_sig("mi355_dense_tile_repack", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i64])
_sig("mi355_dense_tile_index", ctypes.c_int64, [c_i32] * 4)
_sig("mi355_gptq_linear", ctypes.c_int, [c_vp] * 5 + [c_i32, c_i32, c_vp, c_vp] + [c_i32] * 6 + [c_i64])
_sig("mi355_gptq_linear_tiled", ctypes.c_int, [c_vp] * 5 + [c_i32, c_i32, c_vp, c_vp] + [c_i32] * 6 + [c_i64])
_sig("mi355_gptq_tile_repack", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_gptq_tile_unpack", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64])
_sig("mi355_gptq_tile_index", ctypes.c_int64, [c_i32] * 4)


class LlamaConfig(ctypes.Structure):
    """mirror of `mi355_llama_config`"""
    _fields_ = [(n, c_i32) for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "intermediate",
                                     "vocab", "max_seq", "block_size", "kv_layout", "max_batch",
                                     "max_blocks_per_seq")] + \
               [("rms_eps", c_f32), ("rope_theta", c_f32), ("tp_rank", c_i32), ("tp_world", c_i32),
                ("n_expert", c_i32), ("n_expert_used", c_i32)]


_sig("mi355_llama_set_moe_expert", ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32])
_sig("mi355_llama_create", c_vp, [ctypes.POINTER(LlamaConfig)])
_sig("mi355_llama_destroy", None, [c_vp])
_sig("mi355_llama_set_qweight", ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32])
_sig("mi355_llama_set_qweight_tiles", ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32])
_sig("mi355_llama_set_f32", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_i64])
_sig("mi355_llama_alloc_kv_cache", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_llama_kv_ptr", c_vp, [c_vp, c_i32, c_i32])
_sig("mi355_llama_kv_bytes_per_tensor", c_i64, [c_vp])
_sig("mi355_llama_kv_copy", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_i64, c_i32])
_sig("mi355_llama_forward_decode", ctypes.c_int, [c_vp] * 6 + [c_i32, c_i32, c_i32, c_vp, c_i64])
_sig("mi355_llama_forward_prefill", ctypes.c_int, [c_vp] * 7 + [c_i32] * 4 + [c_vp, c_i64])
_sig("mi355_llama_decode_begin", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_llama_set_graph", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_llama_decode_step", ctypes.c_int, [c_vp, c_i64])
_sig("mi355_llama_graph_captures", c_i64, [c_vp])
_sig("mi355_llama_eager_steps", c_i64, [c_vp])
_sig("mi355_llama_decode_read_tokens", ctypes.c_int, [c_vp, c_vp, c_i64])
_sig("mi355_llama_logits_ptr", c_vp, [c_vp])
_sig("mi355_comm_unique_id", ctypes.c_int, [c_vp])
_sig("mi355_llama_init_comm", ctypes.c_int, [c_vp, c_vp])
_sig("mi355_llama_run_part", ctypes.c_int, [c_vp, c_i32, c_i32, c_i64])
_sig("mi355_llama_act_ptr", c_vp, [c_vp, c_i32])


# ---- host block manager (section 5 of the header): every entry point takes plain ints / pointers
c_u32, c_u64 = ctypes.c_uint32, ctypes.c_uint64
_sig("mi355_be_create", c_vp, [c_i32] * 5)
_sig("mi355_be_destroy", None, [c_vp])
for _n in ("mi355_be_num_free_blocks", "mi355_be_num_free_cpu_blocks", "mi355_be_num_blocks",
           "mi355_be_prefix_cache_blocks", "mi355_pc_cached_blocks", "mi355_pc_lru_len"):
    _sig(_n, c_i32, [c_vp])
_sig("mi355_be_free_block_ids", c_i32, [c_vp, c_vp, c_i32])
_sig("mi355_be_seq_create", c_i32, [c_vp, c_i64, c_vp, c_i32])
for _n in ("mi355_be_seq_remove", "mi355_be_seq_len", "mi355_be_seq_logical_blocks", "mi355_be_seq_get_cached_tokens",
           "mi355_be_seq_has_prefix_hash", "mi355_be_pop_back_block", "mi355_be_free_sequence",
           "mi355_be_cache_sequence", "mi355_be_fallback_to_full_prefill"):
    _sig(_n, c_i32, [c_vp, c_i64])
_sig("mi355_be_seq_add_token", c_i32, [c_vp, c_i64, c_u32])
for _n in ("mi355_be_seq_set_cached_tokens", "mi355_be_seq_set_warmup_tokens", "mi355_be_seq_prefill_chunk_tokens",
           "mi355_be_rebuild_with_cached_prefix"):
    _sig(_n, c_i32, [c_vp, c_i64, c_i32])
_sig("mi355_be_block_table", c_i32, [c_vp, c_i64, c_vp, c_i32])
_sig("mi355_be_block_refcount", c_i32, [c_vp, c_i32])
for _n in ("mi355_be_can_allocate", "mi355_be_allocate", "mi355_be_prefill_chunk_blocks_required",
           "mi355_be_can_append_prefill_chunk", "mi355_be_append_prefill_chunk_slots"):
    _sig(_n, c_i32, [c_vp, c_vp, c_i32, c_i32])
for _n in ("mi355_be_can_append_token", "mi355_be_can_swap_out", "mi355_be_swap_in_required_blocks",
           "mi355_be_can_swap_in"):
    _sig(_n, c_i32, [c_vp, c_vp, c_i32])
_sig("mi355_be_append_token_slot", c_i32, [c_vp, c_i64, c_vp, c_vp])
for _n in ("mi355_be_evict_prefix_cache_blocks", "mi355_be_evict_prefix_cache_until_free"):
    _sig(_n, c_i32, [c_vp, c_i32])
_sig("mi355_be_query_prefix_match_tokens", c_i32, [c_vp, c_vp, c_i32])
for _n in ("mi355_be_swap_out", "mi355_be_swap_in"):
    _sig(_n, c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32])
for _n in ("mi355_be_finalize_swap_out", "mi355_be_rollback_swap_out", "mi355_be_finalize_swap_in",
           "mi355_be_rollback_swap_in"):
    _sig(_n, None, [c_vp, c_i64])
_sig("mi355_tuning_supported", c_i32, [c_i32])
_sig("mi355_be_test_refuse_swaps", None, [c_vp, c_i32, c_i32])
_sig("mi355_pc_create", c_vp, [c_i32] * 4)
_sig("mi355_pc_insert", c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32])
_sig("mi355_pc_match", c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32])
_sig("mi355_pc_evict", c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32])
_sig("mi355_pc_hash_for_blocks", c_u64, [c_vp, c_vp, c_i32, c_i32, c_i32, c_u64, c_i32])
_sig("mi355_be_prepare_decode", c_i32, [c_vp, c_vp, c_i32] + [c_vp] * 5 + [c_i32])
_sig("mi355_be_prepare_prompt", c_i32, [c_vp, c_vp, c_i32, c_i32] + [c_vp] * 7 + [c_i32, c_i32, c_vp])

# ---- scheduler (section 6)
_sig("mi355_sched_create", c_vp, [c_i32] * 8)
_sig("mi355_sched_destroy", None, [c_vp])
_sig("mi355_sched_block_engine", c_vp, [c_vp])
_sig("mi355_sched_add_group", c_i32, [c_vp, c_i64, c_vp, c_i32, c_u64])
for _n in ("mi355_sched_group_status", "mi355_sched_set_group_finished"):
    _sig(_n, c_i32, [c_vp, c_i64])
_sig("mi355_sched_queue_len", c_i32, [c_vp, c_i32])
for _n in ("mi355_sched_has_unfinished", "mi355_sched_is_last_prefill"):
    _sig(_n, c_i32, [c_vp])
_sig("mi355_sched_schedule", c_i32, [c_vp, c_u64])
_sig("mi355_sched_result", c_i32, [c_vp, c_i32, c_vp, c_i32])
_sig("mi355_sched_filter_prefill_finished", c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32])
_sig("mi355_sched_free_finished", c_i32, [c_vp, c_vp, c_i32])
_sig("mi355_sched_abort_sequences", c_i32, [c_vp, c_vp, c_i32])
for _n in ("mi355_sched_rollback_swap_in", "mi355_sched_rollback_swap_out"):
    _sig(_n, None, [c_vp, c_i64])


# ---- dense 16-bit model host layer (section 4b)
class DenseConfig(ctypes.Structure):
    """mirror of `mi355_dense_config`"""
    _fields_ = [(n, c_i32) for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "intermediate",
                                     "vocab", "max_seq", "block_size", "kv_layout", "max_batch",
                                     "max_blocks_per_seq")] + \
               [("rms_eps", c_f32), ("rope_theta", c_f32), ("dtype", c_i32), ("rope_interleaved", c_i32),
                ("norm_type", c_i32), ("rotary_dim", c_i32), ("kv_fp8", c_i32), ("tp_rank", c_i32), ("tp_world", c_i32),
                ("vocab_total", c_i32)]


_sig("mi355_layer_norm", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_i64])
_sig("mi355_dense_create", c_vp, [ctypes.POINTER(DenseConfig)])
_sig("mi355_dense_destroy", None, [c_vp])
_sig("mi355_dense_set_weight", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_i64])
_sig("mi355_dense_set_weight_dev", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_i64])
_sig("mi355_dense_set_gptq", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32])
_sig("mi355_dense_set_comm", ctypes.c_int, [c_vp, c_vp])
_sig("mi355_dense_set_rope_tables", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32])
_sig("mi355_llama_set_rope_tables", ctypes.c_int, [c_vp, c_vp, c_vp, c_i32])
_sig("mi355_abi_struct_size", c_i64, [c_i32])
_sig("mi355_rope_table_len", c_i32, [c_vp, c_i32, c_i32])
_sig("mi355_rope_tables", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, ctypes.c_double, c_vp, c_i32, c_i32])
_sig("mi355_comm_create", c_vp, [c_vp, c_i32, c_i32])
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, c_i64, c_i32, c_i64)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64)
_sig("mi355_comm_create_external", c_vp, [ALLREDUCE_FN, ALLGATHER_FN, c_vp])
_sig("mi355_llama_set_comm", ctypes.c_int, [c_vp, c_vp])
_sig("mi355_comm_destroy", None, [c_vp])
_sig("mi355_comm_all_reduce", ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i64])
_sig("mi355_comm_all_gather", ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64])
_sig("mi355_llama_comm_handle", c_vp, [c_vp])
_sig("mi355_comm_set_options", ctypes.c_int, [c_vp, c_i32, c_i32])
_sig("mi355_comm_wire_bf16", ctypes.c_int, [c_vp])
_sig("mi355_comm_all_reduce_residual", ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64])
_sig("mi355_comm_p2p_export", ctypes.c_int, [c_vp, c_vp])
_sig("mi355_comm_p2p_attach", ctypes.c_int, [c_vp, c_vp, c_i32, c_i32])
_sig("mi355_comm_p2p_error", ctypes.c_int, [c_vp])
_sig("mi355_comm_p2p_enable", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_comm_capture_probe", ctypes.c_int, [c_vp, c_i64])
_sig("mi355_dense_alloc_kv_cache", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_dense_kv_ptr", c_vp, [c_vp, c_i32, c_i32])
_sig("mi355_dense_set_layer_window", ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp])
_sig("mi355_dense_forward", ctypes.c_int, [c_vp] * 7 + [c_i32] * 5 + [c_vp, c_i64])
_sig("mi355_llama_set_attention_numerics", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_internal_qmm_set_exact", None, [c_i32])
_sig("mi355_internal_qmm_get_exact", c_i32, [])
_sig("mi355_paged_attention_reference_numerics", ctypes.c_int, [c_vp] * 6 + [c_i32] * 7 + [c_f32, c_i32, c_i64])
_sig("mi355_dense_finalize", ctypes.c_int, [c_vp])
_sig("mi355_dense_set_graph", ctypes.c_int, [c_vp, c_i32])
_sig("mi355_dense_decode_begin", ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64])
_sig("mi355_dense_decode_step", ctypes.c_int, [c_vp, c_i64])
_sig("mi355_dense_graph_captures", c_i64, [c_vp])
_sig("mi355_dense_eager_steps", c_i64, [c_vp])
_sig("mi355_dense_decode_read_tokens", ctypes.c_int, [c_vp, c_vp, c_i64])
_sig("mi355_dense_logits_ptr", c_vp, [c_vp])

# ---- GGUF reader (section 7)
_sig("mi355_gguf_open", c_vp, [ctypes.c_char_p])
_sig("mi355_gguf_close", None, [c_vp])
_sig("mi355_gguf_version", c_i32, [c_vp])
_sig("mi355_gguf_n_tensors", c_i32, [c_vp])
_sig("mi355_gguf_get_u64", c_i32, [c_vp, ctypes.c_char_p, c_vp])
_sig("mi355_gguf_get_f64", c_i32, [c_vp, ctypes.c_char_p, c_vp])
_sig("mi355_gguf_get_str", c_i32, [c_vp, ctypes.c_char_p, c_vp, c_i32])
_sig("mi355_gguf_find", c_i32, [c_vp, ctypes.c_char_p])
_sig("mi355_gguf_tensor_info", c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp])
_sig("mi355_gguf_tensor_data", c_vp, [c_vp, c_i32])
_sig("mi355_llama_load_gguf", ctypes.c_int, [ctypes.c_char_p] + [c_i32] * 5 + [c_vp, c_vp])
_sig("mi355_llama_load_gguf_tp", ctypes.c_int, [ctypes.c_char_p] + [c_i32] * 7 + [c_vp, c_vp])
_sig("mi355_llama_check_gguf", ctypes.c_int, [ctypes.c_char_p, c_i32, c_i32, c_vp])
_sig("mi355_kvcache_budget_bytes", c_i64, [c_i64, c_f32])
_sig("mi355_effective_max_seq_len", c_i64, [c_vp, c_i64])
_sig("mi355_get_cache_config", ctypes.c_int, [c_i64, c_i64] + [c_i32] * 7 + [c_vp, c_vp])
_sig("mi355_gguf_tensor_shard", ctypes.c_int64, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, ctypes.c_int64])
_sig("mi355_gguf_tensor_shard_q8_0", ctypes.c_int64, [c_vp, c_i32, c_i32, c_i32, c_vp, ctypes.c_int64])
_sig("mi355_gguf_tensor_rows_padded", ctypes.c_int64, [c_vp, c_i32, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64])


# the hand-written ctypes mirrors must have the layout the library was built with
for _id, _cls in ((0, QmmDesc), (1, LlamaConfig), (2, DenseConfig), (3, RopeScaling)):
    if lib.mi355_abi_struct_size(_id) != ctypes.sizeof(_cls):
        raise ImportError(f"{_cls.__name__}: ctypes mirror is {ctypes.sizeof(_cls)} bytes, the library expects "
                          f"{lib.mi355_abi_struct_size(_id)} (include/mi355_vllm.h changed?)")
