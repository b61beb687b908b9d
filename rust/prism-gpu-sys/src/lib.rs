//! Raw bindings to libprism_gpu.so: include/prism_gpu.h, one to one.
//!
//! Arrow data crosses as Arrow C Data Interface structs, which arrow-rs produces and
//! consumes as `arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema}` (same layout as
//! include/arrow_c_data.h).
#![allow(non_camel_case_types)]

use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use std::os::raw::{c_char, c_int, c_void};

pub const PRISM_GPU_ABI_VERSION: u32 = 1;
pub const PRISM_GPU_MAX_SLOTS: usize = 24;
pub const PRISM_GPU_MAX_KERNEL_STATS: usize = 24;
pub const PRISM_GPU_COMM_ID_BYTES: usize = 128;

// prism_gpu_status
pub const PRISM_GPU_OK: c_int = 0;
pub const PRISM_GPU_EINVAL: c_int = -1;
pub const PRISM_GPU_EUNSUPPORTED: c_int = -2;
pub const PRISM_GPU_EOOM: c_int = -3;
pub const PRISM_GPU_EHIP: c_int = -4;
pub const PRISM_GPU_ERCCL: c_int = -5;
pub const PRISM_GPU_EINTERNAL: c_int = -6;

// prism_gpu_expr_kind
pub const PRISM_EXPR_COLUMN: i32 = 1;
pub const PRISM_EXPR_LIT_I64: i32 = 2;
pub const PRISM_EXPR_LIT_STR: i32 = 3;
pub const PRISM_EXPR_CMP: i32 = 4;
pub const PRISM_EXPR_AND: i32 = 5;
pub const PRISM_EXPR_OR: i32 = 6;
pub const PRISM_EXPR_NOT: i32 = 7;
pub const PRISM_EXPR_LIKE: i32 = 8;
pub const PRISM_EXPR_IS_NULL: i32 = 9;
pub const PRISM_EXPR_IS_NOT_NULL: i32 = 10;
// prism_gpu_cmp_op
pub const PRISM_CMP_EQ: i32 = 1;
pub const PRISM_CMP_NEQ: i32 = 2;
pub const PRISM_CMP_LT: i32 = 3;
pub const PRISM_CMP_LTEQ: i32 = 4;
pub const PRISM_CMP_GT: i32 = 5;
pub const PRISM_CMP_GTEQ: i32 = 6;
pub const PRISM_LIKE_NEGATED: i32 = 1;
pub const PRISM_LIKE_CASE_INSENSITIVE: i32 = 2;
// prism_gpu_key_kind
pub const PRISM_KEY_COLUMN: i32 = 1;
pub const PRISM_KEY_DATE_BIN: i32 = 2;
// prism_gpu_agg_func
pub const PRISM_AGG_COUNT_STAR: i32 = 1;
pub const PRISM_AGG_COUNT: i32 = 2;
pub const PRISM_AGG_SUM: i32 = 3;
pub const PRISM_AGG_MIN: i32 = 4;
pub const PRISM_AGG_MAX: i32 = 5;
pub const PRISM_AGG_AVG: i32 = 6;
pub const PRISM_AGG_COUNT_DISTINCT: i32 = 7;
// prism_gpu_agg_mode
pub const PRISM_MODE_SINGLE: i32 = 0;
pub const PRISM_MODE_PARTIAL: i32 = 1;
// prism_gpu_reduce_op / prism_gpu_reduce_type
pub const PRISM_REDUCE_SUM: c_int = 0;
pub const PRISM_REDUCE_MIN: c_int = 1;
pub const PRISM_REDUCE_MAX: c_int = 2;
pub const PRISM_REDUCE_I64: c_int = 0;
pub const PRISM_REDUCE_F64: c_int = 1;

// opaque handles
#[repr(C)] pub struct prism_gpu_ctx { _p: [u8; 0] }
#[repr(C)] pub struct prism_gpu_table { _p: [u8; 0] }
#[repr(C)] pub struct prism_gpu_plan { _p: [u8; 0] }
#[repr(C)] pub struct prism_gpu_part { _p: [u8; 0] }
#[repr(C)] pub struct prism_gpu_state { _p: [u8; 0] }
#[repr(C)] pub struct prism_gpu_comm { _p: [u8; 0] }

#[repr(C)]
pub struct prism_gpu_expr {
    pub kind: i32,
    pub op: i32,
    pub lhs: i32,
    pub rhs: i32,
    pub column: i32,
    pub flags: i32,
    pub i64: i64,
    pub str: *const c_char,
    pub str_len: u64,
}

#[repr(C)]
pub struct prism_gpu_key {
    pub kind: i32,
    pub column: i32,
    pub stride_ms: i64,
    pub origin_ms: i64,
    pub name: *const c_char,
}

#[repr(C)]
pub struct prism_gpu_agg {
    pub func: i32,
    pub column: i32,
    pub name: *const c_char,
}

#[repr(C)]
pub struct prism_gpu_plan_desc {
    pub abi_version: u32,
    pub mode: i32,
    pub exprs: *const prism_gpu_expr,
    pub n_exprs: u32,
    pub predicate: i32,
    pub keys: *const prism_gpu_key,
    pub n_keys: u32,
    pub aggs: *const prism_gpu_agg,
    pub n_aggs: u32,
    pub order_by: i32,
    pub descending: i32,
    pub limit: i64,
    pub groups_hint: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct prism_gpu_col_stats {
    pub min: i64,
    pub max: i64,
    pub nulls: u64,
}

#[repr(C)]
pub struct prism_gpu_state_info {
    pub dense: i32,
    pub n_slots: i32,
    pub entries: u64,
    pub export_bytes: u64,
    pub slot_merge: [i32; PRISM_GPU_MAX_SLOTS],
}

#[repr(C)]
pub struct prism_gpu_transport {
    pub user: *mut c_void,
    pub allgather: Option<unsafe extern "C" fn(user: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: usize) -> c_int>,
    pub reduce: Option<unsafe extern "C" fn(user: *mut c_void, buf: *mut c_void, n: usize, type_: c_int, op: c_int, root: c_int) -> c_int>,
    pub alltoallv: Option<unsafe extern "C" fn(user: *mut c_void, send: *const u64, send_counts: *const u64, recv: *mut u64, recv_counts: *const u64) -> c_int>,
}

#[repr(C)]
pub struct prism_gpu_comm_stats {
    pub collectives: u64,
    pub bytes_sent: u64,
    pub exchange_ms: f64,
}

#[repr(C)]
pub struct prism_gpu_kernel_stat {
    pub name: [c_char; 48],
    pub launches: u64,
    pub total_ms: f64,
    pub algorithmic_bytes: u64,
}

#[repr(C)]
pub struct prism_gpu_stats {
    pub rows_scanned: u64,
    pub rows_selected: u64,
    pub groups_out: u64,
    pub bytes_staged: u64,
    pub staging_ms: f64,
    pub n_kernels: u32,
    pub kernels: [prism_gpu_kernel_stat; PRISM_GPU_MAX_KERNEL_STATS],
    pub chunks_pruned: u64,
}

extern "C" {
    // ---- context ----
    pub fn prism_gpu_init(device_ordinal: c_int, out: *mut *mut prism_gpu_ctx) -> c_int;
    pub fn prism_gpu_ctx_free(ctx: *mut prism_gpu_ctx);
    pub fn prism_gpu_set_option(ctx: *mut prism_gpu_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn prism_gpu_last_error() -> *const c_char;

    // ---- columnar chunk loader ----
    pub fn prism_gpu_table_create(ctx: *mut prism_gpu_ctx, schema: *const FFI_ArrowSchema, out: *mut *mut prism_gpu_table) -> c_int;
    pub fn prism_gpu_table_append(table: *mut prism_gpu_table, partition_id: u32, batch: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_table_num_rows(table: *const prism_gpu_table, out: *mut u64) -> c_int;
    pub fn prism_gpu_table_num_partitions(table: *const prism_gpu_table, out: *mut u32) -> c_int;
    pub fn prism_gpu_table_free(table: *mut prism_gpu_table);
    pub fn prism_gpu_table_append_parquet(table: *mut prism_gpu_table, partition_id: u32, file: *const c_void, file_len: usize) -> c_int;
    pub fn prism_gpu_parquet_read(file: *const c_void, file_len: usize, out_schema: *mut FFI_ArrowSchema, out: *mut FFI_ArrowArray) -> c_int;

    // ---- plans ----
    pub fn prism_gpu_plan_create(ctx: *mut prism_gpu_ctx, input_schema: *const FFI_ArrowSchema, desc: *const prism_gpu_plan_desc, out: *mut *mut prism_gpu_plan) -> c_int;
    pub fn prism_gpu_plan_output_schema(plan: *const prism_gpu_plan, out: *mut FFI_ArrowSchema) -> c_int;
    pub fn prism_gpu_plan_time_range(plan: *const prism_gpu_plan, column: i32, start_ms: *mut i64, end_ms: *mut i64) -> c_int;
    pub fn prism_gpu_plan_free(plan: *mut prism_gpu_plan);
    pub fn prism_gpu_execute(plan: *mut prism_gpu_plan, table: *mut prism_gpu_table, partition_ids: *const u32, n_partitions: usize, out: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_execute_noalloc(plan: *mut prism_gpu_plan, table: *mut prism_gpu_table, partition_ids: *const u32, n_partitions: usize, out_values: *mut i64, out_valid: *mut u8, out_stride: u64, n_groups: *mut u64) -> c_int;

    // ---- streaming form ----
    pub fn prism_gpu_partition_open(plan: *mut prism_gpu_plan, partition: u32, out: *mut *mut prism_gpu_part) -> c_int;
    pub fn prism_gpu_push_batch(part: *mut prism_gpu_part, batch: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_finish(part: *mut prism_gpu_part, out: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_part_free(part: *mut prism_gpu_part);
    pub fn prism_gpu_merge(plan: *mut prism_gpu_plan, states: *const *mut FFI_ArrowArray, n_states: usize, out: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_plan_final_schema(plan: *const prism_gpu_plan, out: *mut FFI_ArrowSchema) -> c_int;

    // ---- device-resident aggregate state ----
    pub fn prism_gpu_scan_stats(plan: *mut prism_gpu_plan, table: *mut prism_gpu_table, partition_ids: *const u32, n_partitions: usize, out: *mut prism_gpu_col_stats) -> c_int;
    pub fn prism_gpu_aggregate(plan: *mut prism_gpu_plan, table: *mut prism_gpu_table, partition_ids: *const u32, n_partitions: usize, stats: *const prism_gpu_col_stats, out: *mut *mut prism_gpu_state) -> c_int;
    pub fn prism_gpu_state_describe(state: *mut prism_gpu_state, out: *mut prism_gpu_state_info) -> c_int;
    pub fn prism_gpu_state_export(state: *mut prism_gpu_state, device_dst: *mut c_void, bytes: usize) -> c_int;
    pub fn prism_gpu_state_import(state: *mut prism_gpu_state, device_src: *const c_void, bytes: usize) -> c_int;
    pub fn prism_gpu_state_merge(state: *mut prism_gpu_state, device_src: *const c_void, n_groups: u64) -> c_int;
    pub fn prism_gpu_state_finish(state: *mut prism_gpu_state, out: *mut FFI_ArrowArray) -> c_int;
    pub fn prism_gpu_state_free(state: *mut prism_gpu_state);
    pub fn prism_gpu_table_dictionary_digest(table: *mut prism_gpu_table, column: i32, digest: *mut u64, n_values: *mut u32) -> c_int;

    // ---- the exchange inside the library ----
    pub fn prism_gpu_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn prism_gpu_comm_init(ctx: *mut prism_gpu_ctx, unique_id: *const c_void, rank: c_int, world: c_int, out: *mut *mut prism_gpu_comm) -> c_int;
    pub fn prism_gpu_comm_init_callbacks(ctx: *mut prism_gpu_ctx, transport: *const prism_gpu_transport, rank: c_int, world: c_int, out: *mut *mut prism_gpu_comm) -> c_int;
    pub fn prism_gpu_comm_free(comm: *mut prism_gpu_comm);
    pub fn prism_gpu_execute_sharded(comm: *mut prism_gpu_comm, plan: *mut prism_gpu_plan, table: *mut prism_gpu_table, partition_ids: *const u32, n_partitions: usize, root: c_int, out: *mut FFI_ArrowArray, has_result: *mut c_int) -> c_int;
    pub fn prism_gpu_comm_get_stats(comm: *mut prism_gpu_comm, out: *mut prism_gpu_comm_stats) -> c_int;

    // ---- measurement ----
    pub fn prism_gpu_set_profiling(ctx: *mut prism_gpu_ctx, enabled: c_int) -> c_int;
    pub fn prism_gpu_get_stats(ctx: *mut prism_gpu_ctx, out: *mut prism_gpu_stats) -> c_int;
    pub fn prism_gpu_reset_stats(ctx: *mut prism_gpu_ctx) -> c_int;
}
