"""ctypes binding of libvsgpu.so (include/vsgpu.h).  Product path: there is no CPU fallback — if the shared library
is missing or no HIP device is present, calls raise VsError."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_LIB_PATH") or os.path.join(_HERE, "libvsgpu.so")  # (VS_LIB_PATH: A/B sessions time another build of the same ABI, scripts/ab_bench_libs.sh)

VS_INVALID_NODE = 0xFFFFFFFF
VS_COSINE, VS_L2, VS_IP = 0, 1, 2
VS_STORAGE_SBQ, VS_STORAGE_PLAIN = 0, 1
ARR_CODES, ARR_NBRS, ARR_TIDS, ARR_VECS, ARR_MEAN, ARR_M2, ARR_VNORM, ARR_LABEL_OFF, ARR_LABEL_VAL = range(9)


class VsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvsgpu error {code}: {msg}")
        self.code = code


class IndexDesc(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("n", "dim_full", "dim_index", "bits", "words", "num_neighbors",
                                          "distance_type", "has_labels", "default_start", "n_label_starts", "storage_type")]


class IndexHost(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("nbrs", C.c_void_p), ("nbr_stride", C.c_uint32), ("heap_tids", C.c_void_p),
                ("vecs", C.c_void_p), ("mean", C.c_void_p), ("m2", C.c_void_p), ("count", C.c_uint64),
                ("label_off", C.c_void_p), ("label_val", C.c_void_p), ("label_start_labels", C.c_void_p),
                ("label_start_nodes", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("queries", "visited_nodes", "candidate_nodes",
                                          "quantized_distance_comparisons", "full_distance_comparisons",
                                          "node_reads", "node_heap_reads", "next_calls", "retries",
                                          "fallback_scans", "fallback_visited_nodes",
                                          "fallback_quantized_distance_comparisons")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TuneEntry(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("step_ms", C.c_float), ("search_ms", C.c_float), ("applicable", C.c_uint32),
                ("rows_identical", C.c_uint32), ("chosen", C.c_uint32), ("error", C.c_int32)]

    def as_dict(self):
        return {"name": self.name.decode(), "step_ms": round(float(self.step_ms), 4), "search_ms": round(float(self.search_ms), 4),
                "applicable": bool(self.applicable), "rows_identical": bool(self.rows_identical), "chosen": bool(self.chosen),
                "error": int(self.error)}


class NodeLayout(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("root_size", "off_heap_item_pointer", "off_bq_vector",
                                          "off_neighbor_index_pointers", "off_labels")]


class MetaLayout(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("root_size", "off_magic_number", "off_version", "off_extension_version_when_built",
                                          "off_distance_type", "off_num_dimensions", "off_num_dimensions_to_index",
                                          "off_bq_num_bits_per_dimension", "off_storage_type", "off_num_neighbors",
                                          "off_search_list_size", "off_max_alpha", "off_start_nodes", "off_quantizer_metadata",
                                          "off_has_labels")]


class MetaPage(C.Structure):
    _fields_ = [("magic_number", C.c_uint32), ("version", C.c_uint32), ("extension_version_when_built", C.c_char * 64),
                ("distance_type", C.c_uint32), ("num_dimensions", C.c_uint32), ("num_dimensions_to_index", C.c_uint32),
                ("bq_num_bits_per_dimension", C.c_uint32), ("storage_type", C.c_uint32), ("num_neighbors", C.c_uint32),
                ("search_list_size", C.c_uint32), ("max_alpha", C.c_double), ("has_start_nodes", C.c_uint32),
                ("default_start_block", C.c_uint32), ("default_start_offset", C.c_uint32), ("n_labeled_start_nodes", C.c_uint32),
                ("quantizer_block", C.c_uint32), ("quantizer_offset", C.c_uint32), ("has_labels", C.c_uint32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["extension_version_when_built"] = d["extension_version_when_built"].decode()
        return d


class HeapAttr(C.Structure):
    _fields_ = [("attlen", C.c_int16), ("attalign", C.c_char)]


class HeapInfo(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("n_nodes", "heap_blocks", "toast_blocks", "n_inline", "n_external", "n_deleted", "n_null",
                                          "n_dead_line_pointer", "n_not_found", "n_toast_incomplete")] + [("n_chunks", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PagesInfo(C.Structure):
    _fields_ = [("n_blocks", C.c_uint32), ("n_nodes", C.c_uint32), ("words", C.c_uint32), ("num_neighbors", C.c_uint32),
                ("has_labels", C.c_uint32), ("n_deleted", C.c_uint32), ("n_label_vals", C.c_uint64),
                ("pages_by_type", C.c_uint32 * 9), ("new_pages", C.c_uint32), ("meta_magic", C.c_uint32),
                ("meta_version", C.c_uint32)]


class BrokerConfig(C.Structure):
    _fields_ = [("max_batch", C.c_uint32), ("max_wait_us", C.c_uint32), ("cursor_lanes", C.c_uint32), ("cursor_pool", C.c_uint32)]


class BrokerStats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("scans", C.c_uint64), ("max_batch", C.c_uint64), ("tasks", C.c_uint64), ("cursors", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8)]


class DatagenParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("dim", C.c_uint32), ("latent_dim", C.c_uint32), ("n_clusters", C.c_uint32),
                ("intra_pct", C.c_uint32), ("noise_pct", C.c_uint32), ("normalize", C.c_uint32)]


# every symbol include/vsgpu.h declares: name -> (restype, argtypes)
_vp, _u32, _u64, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
SYMBOLS = {
    "vs_last_error": (C.c_char_p, []),
    "vs_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "vs_get_option": (_i, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "vs_version": (C.c_char_p, []),
    "vs_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "vs_ctx_create_staging": (_i, [_i, _sz, C.POINTER(_vp)]),
    "vs_ctx_destroy": (None, [_vp]),
    "vs_ctx_sync": (_i, [_vp]),
    "vs_ctx_stream": (_vp, [_vp]),
    "vs_ctx_device_name": (_i, [_vp, C.c_char_p, _sz]),
    "vs_ctx_mem_info": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "vs_profile_enable": (_i, [_vp, _i]),
    "vs_profile_read": (_i, [_vp, C.POINTER(Profile), _i]),
    "vs_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vs_dev_free": (_i, [_vp, _vp]),
    "vs_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "vs_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "vs_index_upload": (_i, [_vp, C.POINTER(IndexDesc), C.POINTER(IndexHost), C.POINTER(_vp)]),
    "vs_index_alloc": (_i, [_vp, C.POINTER(IndexDesc), _i, C.POINTER(_vp)]),
    "vs_index_free": (None, [_vp]),
    "vs_index_view": (_i, [_vp, _vp, _vp]),
    "vs_scanpool_create": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "vs_scanpool_free": (None, [_vp]),
    "vs_scanpool_rescan": (_i, [_vp, _u32, _vp, _vp, _u32, _i]),
    "vs_scanpool_endscan": (_i, [_vp, _u32]),
    "vs_scanpool_fetch": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "vs_scanpool_get_stats": (_i, [_vp, _u32, _vp]),
    "vs_scanpool_get_work": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "vs_index_set_slab": (_i, [_vp, _vp, C.c_size_t]),
    "vs_ws_probe": (_i, [_vp, _vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_float)]),
    "vs_ws_probe_mix": (_i, [_vp, _vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_float)]),
    "vs_index_prepare_workspace": (_i, [_vp]),
    "vs_index_get_desc": (_i, [_vp, C.POINTER(IndexDesc)]),
    "vs_index_array": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_u32)]),
    "vs_index_set_quantizer": (_i, [_vp, _vp, _vp, _u64]),
    "vs_index_set_start_nodes": (_i, [_vp, _u32, _vp, _vp, _u32]),
    "vs_index_set_labels": (_i, [_vp, _vp, _vp]),
    "vs_index_set_visibility": (_i, [_vp, _vp]),
    "vs_index_set_visibility_dev": (_i, [_vp, _vp]),
    "vs_index_get_quantizer": (_i, [_vp, _vp, _vp, C.POINTER(_u64)]),
    "vs_index_download": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _u32]),
    "vs_index_refresh_norms": (_i, [_vp]),
    "vs_index_mark_deleted": (_i, [_vp, _vp, _u32]),
    "vs_node_layout_default": (_i, [_i, C.POINTER(NodeLayout)]),
    "vs_pages_open": (_i, [_u32, _i, C.POINTER(NodeLayout), _u32, C.POINTER(_vp)]),
    "vs_pages_open_plain": (_i, [_u32, C.POINTER(NodeLayout), _u32, C.POINTER(_vp)]),
    "vs_plain_layout_default": (_i, [C.POINTER(NodeLayout)]),
    "vs_pages_add": (_i, [_vp, _u32, _vp, _u32]),
    "vs_pages_finish": (_i, [_vp, C.POINTER(PagesInfo)]),
    "vs_pages_host": (_i, [_vp, C.POINTER(IndexHost)]),
    "vs_pages_node_of": (_i, [_vp, _u32, _u32, C.POINTER(_u32)]),
    "vs_pages_item_pointer_of": (_i, [_vp, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "vs_pages_read_chain": (_i, [_vp, _u32, _u32, _i, _vp, _sz, C.POINTER(_sz)]),
    "vs_pages_sbq_means": (_i, [_vp, _u32, _u32, _vp, _vp, _u32, C.POINTER(_u32), C.POINTER(_u64)]),
    "vs_pages_close": (None, [_vp]),
    "vs_heap_open": (_i, [_u32, C.POINTER(HeapAttr), _u32, _u32, _u32, _vp, _u32, _vp, _u32, C.POINTER(_vp)]),
    "vs_heap_add": (_i, [_vp, _u32, _vp, _u32]),
    "vs_heap_toast_add": (_i, [_vp, _u32, _vp, _u32]),
    "vs_heap_finish": (_i, [_vp, C.POINTER(HeapInfo), _vp]),
    "vs_heap_close": (None, [_vp]),
    "vs_meta_layout_default": (_i, [C.POINTER(MetaLayout)]),
    "vs_meta_page_decode": (_i, [_vp, _sz, C.POINTER(MetaLayout), C.POINTER(MetaPage), _vp, _vp, _vp, _u32]),
    "vs_pages_meta": (_i, [_vp, C.POINTER(MetaLayout), C.POINTER(MetaPage), C.POINTER(IndexDesc), _vp, _vp, _u32]),
    "vs_pages_dev_meta": (_i, [_vp, C.POINTER(MetaLayout), C.POINTER(MetaPage), C.POINTER(IndexDesc), _vp, _vp, _u32]),
    "vs_pages_headers_only": (_i, [_vp]),
    "vs_pages_block_table": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u32)]),
    "vs_pages_dev_open": (_i, [_vp, _u32, C.POINTER(NodeLayout), _u32, C.POINTER(_vp)]),
    "vs_pages_dev_add": (_i, [_vp, _u32, _vp, _u32]),
    "vs_pages_dev_node_of": (_i, [_vp, _u32, _u32, C.POINTER(_u32)]),
    "vs_pages_dev_sbq_means": (_i, [_vp, _u32, _u32, _vp, _vp, _u32, C.POINTER(_u32), C.POINTER(_u64)]),
    "vs_pages_dev_build": (_i, [_vp, C.POINTER(IndexDesc), C.POINTER(IndexHost), C.POINTER(PagesInfo), C.POINTER(_vp)]),
    "vs_pages_dev_close": (None, [_vp]),
    "vs_quantize": (_i, [_vp, _vp, _u32, _vp]),
    "vs_hamming_gather": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "vs_rerank": (_i, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "vs_scan_topk": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "vs_scan_topk_filtered": (_i, [_vp, _vp, _vp, _vp, _i, _u32, _u32, _vp, _vp]),
    "vs_search_batch": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(Stats)]),
    "vs_stream_batch": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, C.POINTER(Stats)]),
    "vs_search_batch_dev": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vs_search_batch_dev_finish": (_i, [_vp, C.POINTER(Stats)]),
    "vs_index_autotune": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, C.c_char_p, C.POINTER(TuneEntry), _u32, C.POINTER(_u32)]),
    "vs_index_set_variant": (_i, [_vp, C.c_char_p]),
    "vs_index_get_variant": (_i, [_vp, C.c_char_p, C.c_size_t]),
    "vs_beginscan": (_i, [_vp, C.POINTER(_vp)]),
    "vs_rescan": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _u32]),
    "vs_gettuple": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u32), C.POINTER(C.c_float)]),
    "vs_scan_xs_recheck": (_i, [_vp]),
    "vs_scan_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "vs_scan_get_work": (_i, [_vp, C.POINTER(Stats), C.POINTER(_u32)]),
    "vs_endscan": (None, [_vp]),
    "vs_broker_create": (_i, [_vp, C.POINTER(BrokerConfig), C.POINTER(_vp)]),
    "vs_broker_search": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vs_broker_search_snapshot": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vs_broker_snapshot_put": (_i, [_vp, _u32, _vp]),
    "vs_index_snapshot_put": (_i, [_vp, _u32, _vp]),
    "vs_index_has_neighbor_masks": (_i, [_vp]),
    "vs_index_snapshot_use": (_i, [_vp, _u32, _vp]),
    "vs_index_snapshot_share": (_i, [_vp, _vp]),
    "vs_index_device": (_i, [_vp]),
    "vs_broker_get_stats": (_i, [_vp, C.POINTER(BrokerStats)]),
    "vs_broker_destroy": (None, [_vp]),
    "vs_broker_index": (_vp, [_vp]),
    "vs_beginscan_on_broker": (_i, [_vp, C.POINTER(_vp)]),
    "vs_broker_call": (_i, [_vp, _vp, _vp]),
    "vs_scan_set_snapshot": (_i, [_vp, _u32]),
    "vs_scan_prefetch": (_i, [_vp, _u32]),
    "vs_shm_client_fetch": (_i, [_vp, C.c_uint64, _vp, _vp, _u32, _i, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "vs_shm_client_end_scan": (_i, [_vp, C.c_uint64]),
    "vs_shm_server_create": (_i, [_vp, C.c_char_p, _u32, _u32, C.POINTER(BrokerConfig), C.POINTER(_vp)]),
    "vs_shm_server_get_stats": (_i, [_vp, C.POINTER(BrokerStats)]),
    "vs_shm_server_pool_stats": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "vs_shm_server_destroy": (None, [_vp]),
    "vs_shm_server_snapshot_put": (_i, [_vp, _u32, _vp]),
    "vs_shm_client_search_snapshot": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vs_shm_client_open": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "vs_shm_client_dim": (_u32, [_vp]),
    "vs_shm_client_search": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vs_shm_client_close": (None, [_vp]),
    "vs_shard_range": (_i, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "vs_index_replicate": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "vs_multi_create": (_i, [_vp, C.POINTER(_i), _u32, _u32, C.POINTER(_vp)]),
    "vs_multi_size": (_u32, [_vp]),
    "vs_multi_index": (_vp, [_vp, _u32]),
    "vs_multi_ctx": (_vp, [_vp, _u32]),
    "vs_multi_search_batch": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, C.POINTER(Stats)]),
    "vs_multi_stream_batch": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, C.POINTER(Stats)]),
    "vs_multi_destroy": (None, [_vp]),
    "vs_comm_unique_id": (_i, [_vp]),
    "vs_comm_create": (_i, [_vp, _vp, _u32, _u32, C.POINTER(_vp)]),
    "vs_comm_rank": (_u32, [_vp]),
    "vs_comm_world": (_u32, [_vp]),
    "vs_comm_gather_topk": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    "vs_comm_bcast": (_i, [_vp, _vp, _sz, _u32]),
    "vs_comm_replicate_index": (_i, [_vp, _vp, _u32]),
    "vs_comm_destroy": (None, [_vp]),
    "vs_sbq_train": (_i, [_vp]),
    "vs_sbq_quantize_corpus": (_i, [_vp]),
    "vs_build_graph": (_i, [_vp, _u32, C.c_double, _u32, _u64]),
    "vs_index_build_unreachable": (_u32, [_vp]),
    "vs_datagen_fill": (_i, [_vp, C.POINTER(DatagenParams), _u64, _u64, _vp]),
    "vs_bruteforce_topk": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
}

_lib = None


def load():
    """Load libvsgpu.so. torch (if it is going to be used in this process) must be imported BEFORE this so both share
    one HIP runtime (torch bundles its own libamdhip64 with the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VsError(-2, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if not os.environ.get("VS_NO_TORCH"):  # (VS_NO_TORCH=1: a process that will never import torch skips the seconds its import costs)
        try:
            import torch  # noqa: F401  (pins the HIP runtime copy when torch is installed)
        except Exception:  # pragma: no cover
            pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if os.environ.get("VS_LIB_TOLERANT") and not hasattr(L, name):
            continue  # (bisecting with a library of an older commit: scripts/fuzz_emu.py --lib)
        f = getattr(L, name)  # AttributeError if the ABI lost a symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(code):
    if code < 0:
        raise VsError(code, load().vs_last_error().decode("utf-8", "replace"))
    return code
