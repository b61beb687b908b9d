//! DataFusion operator over libprism_gpu.so for prism-query.
//!
//! `GpuOffloadRule` is registered where the reference builds its session
//! (src/prism-query/src/main.rs:82, `SessionContext::new()`):
//!
//! ```ignore
//! let gpu = Arc::new(Gpu::new(0)?);
//! let state = SessionState::with_config_rt(SessionConfig::new(), Arc::new(RuntimeEnv::default()))
//!     .add_physical_optimizer_rule(Arc::new(GpuOffloadRule::new(gpu)));
//! let ctx = SessionContext::with_state(state);
//! ```
//!
//! It rewrites `AggregateExec(Partial) <- [CoalesceBatchesExec] <- [FilterExec] <- scan` (the scan is
//! the ParquetExec of src/prism-query/src/providers/table_provider.rs:130) into one `GpuAggExec`;
//! DataFusion's own `AggregateExec(Final)` above it consumes the PARTIAL state batches unchanged.
//! A plan the library refuses (`PRISM_GPU_EUNSUPPORTED`) keeps DataFusion's operators.

use std::any::Any;
use std::ffi::{CStr, CString};
use std::fmt;
use std::os::raw::c_int;
use std::sync::Arc;

use arrow::array::{Array, StructArray};
use arrow::datatypes::{DataType, SchemaRef, TimeUnit};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use arrow::record_batch::RecordBatch;
use datafusion::common::{DataFusionError, Result, ScalarValue, Statistics};
use datafusion::config::ConfigOptions;
use datafusion::execution::TaskContext;
use datafusion::logical_expr::Operator;
use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::coalesce_batches::CoalesceBatchesExec;
use datafusion::physical_plan::expressions::{
    Avg, BinaryExpr, Column, Count, DistinctCount, IsNotNullExpr, IsNullExpr, LikeExpr, Literal, Max, Min, NotExpr, Sum,
};
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{
    AggregateExpr, DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PhysicalExpr, SendableRecordBatchStream,
};
use datafusion::physical_expr::{PhysicalSortExpr, ScalarFunctionExpr};
use futures::{StreamExt, TryStreamExt};
use prism_gpu_sys as sys;

/// The error convention of the reference on this path (table_provider.rs:85,93 map failures to
/// DataFusionError::Internal / ::Execution).
fn check(rc: c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::prism_gpu_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        sys::PRISM_GPU_EUNSUPPORTED => DataFusionError::NotImplemented(msg),
        sys::PRISM_GPU_EOOM => DataFusionError::ResourcesExhausted(msg),
        sys::PRISM_GPU_EINTERNAL => DataFusionError::Internal(msg),
        _ => DataFusionError::Execution(msg),
    })
}

/// One GPU of the process (prism_gpu_ctx).
pub struct Gpu(*mut sys::prism_gpu_ctx);
unsafe impl Send for Gpu {}
unsafe impl Sync for Gpu {} // the library locks what contexts share between threads

impl Gpu {
    pub fn new(device: i32) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::prism_gpu_init(device, &mut ctx) })?;
        Ok(Gpu(ctx))
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { sys::prism_gpu_ctx_free(self.0) }
    }
}

/// A compiled plan (prism_gpu_plan); shared by the partitions that execute it.
pub struct GpuPlan {
    raw: *mut sys::prism_gpu_plan,
    output: FFI_ArrowSchema,
    _gpu: Arc<Gpu>,
}
unsafe impl Send for GpuPlan {}
unsafe impl Sync for GpuPlan {}

impl Drop for GpuPlan {
    fn drop(&mut self) {
        unsafe { sys::prism_gpu_plan_free(self.raw) }
    }
}

struct PartGuard(*mut sys::prism_gpu_part);
impl Drop for PartGuard {
    fn drop(&mut self) {
        unsafe { sys::prism_gpu_part_free(self.0) }
    }
}
unsafe impl Send for PartGuard {}

/// Scan -> filter -> AggregateExec(Partial) on the GPU.
pub struct GpuAggExec {
    input: Arc<dyn ExecutionPlan>,
    plan: Arc<GpuPlan>,
    schema: SchemaRef,
}

impl fmt::Debug for GpuAggExec {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "GpuAggExec")
    }
}

impl DisplayAs for GpuAggExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "GpuAggExec: mode=Partial (libprism_gpu)")
    }
}

impl ExecutionPlan for GpuAggExec {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn output_partitioning(&self) -> Partitioning {
        // one PARTIAL state batch per input partition, as AggregateExec(Partial)
        Partitioning::UnknownPartitioning(self.input.output_partitioning().partition_count())
    }
    fn output_ordering(&self) -> Option<&[PhysicalSortExpr]> {
        None
    }
    fn children(&self) -> Vec<Arc<dyn ExecutionPlan>> {
        vec![self.input.clone()]
    }
    fn with_new_children(self: Arc<Self>, children: Vec<Arc<dyn ExecutionPlan>>) -> Result<Arc<dyn ExecutionPlan>> {
        Ok(Arc::new(GpuAggExec { input: children[0].clone(), plan: self.plan.clone(), schema: self.schema.clone() }))
    }
    fn statistics(&self) -> Statistics {
        Statistics::default()
    }

    /// One DataFusion partition = one prism_gpu_part: the input stream is pushed batch by batch
    /// (the library gathers them into ~1M-row chunks, copies each through page-locked memory on
    /// the partition's own stream and aggregates it while the next one gathers), and the
    /// partition's PARTIAL state comes back from prism_gpu_finish.
    fn execute(&self, partition: usize, context: Arc<TaskContext>) -> Result<SendableRecordBatchStream> {
        let mut input = self.input.execute(partition, context)?;
        let plan = self.plan.clone();
        let schema = self.schema.clone();
        let fut = async move {
            let mut part = std::ptr::null_mut();
            check(unsafe { sys::prism_gpu_partition_open(plan.raw, partition as u32, &mut part) })?;
            let guard = PartGuard(part);
            while let Some(batch) = input.next().await {
                let data = StructArray::from(batch?).into_data();
                let (mut array, _schema) = to_ffi(&data)?;
                // moved: the library calls `release` once the rows are on the device
                check(unsafe { sys::prism_gpu_push_batch(guard.0, &mut array) })?;
                std::mem::forget(array); // its release callback is now NULL; nothing left to drop
            }
            // the calls that wait for the device run off the async executor
            tokio::task::spawn_blocking(move || -> Result<RecordBatch> {
                let mut out = FFI_ArrowArray::empty();
                check(unsafe { sys::prism_gpu_finish(guard.0, &mut out) })?;
                let data = from_ffi(out, &plan.output)?;
                Ok(RecordBatch::from(StructArray::from(data)))
            })
            .await
            .map_err(|e| DataFusionError::Execution(e.to_string()))?
        };
        Ok(Box::pin(RecordBatchStreamAdapter::new(schema, futures::stream::once(fut))))
    }
}

/// Replaces AggregateExec(Partial) over a filtered scan with GpuAggExec.
pub struct GpuOffloadRule {
    gpu: Arc<Gpu>,
}

impl GpuOffloadRule {
    pub fn new(gpu: Arc<Gpu>) -> Self {
        GpuOffloadRule { gpu }
    }

    fn try_offload(&self, agg: &AggregateExec) -> Result<Option<Arc<dyn ExecutionPlan>>> {
        if *agg.mode() != AggregateMode::Partial || agg.filter_expr().iter().any(|f| f.is_some()) {
            return Ok(None);
        }
        // AggregateExec(Partial) <- [CoalesceBatchesExec] <- [FilterExec] <- scan
        let mut child = agg.input().clone();
        if let Some(c) = child.as_any().downcast_ref::<CoalesceBatchesExec>() {
            child = c.input().clone();
        }
        let mut predicate: Option<Arc<dyn PhysicalExpr>> = None;
        if let Some(f) = child.as_any().downcast_ref::<FilterExec>() {
            predicate = Some(f.predicate().clone());
            child = f.input().clone();
        }
        let input_schema = child.schema();

        let mut b = DescBuilder::default();
        let root = match &predicate {
            Some(p) => match b.expr(p, &input_schema) {
                Some(i) => i,
                None => return Ok(None),
            },
            None => -1,
        };
        for (e, name) in agg.group_expr().expr() {
            if !b.key(e, name) {
                return Ok(None);
            }
        }
        for a in agg.aggr_expr() {
            if !b.agg(a) {
                return Ok(None);
            }
        }
        let desc = sys::prism_gpu_plan_desc {
            abi_version: sys::PRISM_GPU_ABI_VERSION,
            mode: sys::PRISM_MODE_PARTIAL,
            exprs: b.exprs.as_ptr(),
            n_exprs: b.exprs.len() as u32,
            predicate: root,
            keys: b.keys.as_ptr(),
            n_keys: b.keys.len() as u32,
            aggs: b.aggs.as_ptr(),
            n_aggs: b.aggs.len() as u32,
            order_by: -1,
            descending: 0,
            limit: -1,
            groups_hint: 0,
        };
        let ffi_schema = FFI_ArrowSchema::try_from(input_schema.as_ref())?;
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::prism_gpu_plan_create(self.gpu.0, &ffi_schema, &desc, &mut raw) };
        if rc == sys::PRISM_GPU_EUNSUPPORTED {
            return Ok(None); // keep DataFusion's operators for this plan
        }
        check(rc)?;
        let mut output = FFI_ArrowSchema::empty();
        check(unsafe { sys::prism_gpu_plan_output_schema(raw, &mut output) })?;
        let plan = Arc::new(GpuPlan { raw, output, _gpu: self.gpu.clone() });
        Ok(Some(Arc::new(GpuAggExec { input: child, plan, schema: agg.schema() })))
    }
}

impl PhysicalOptimizerRule for GpuOffloadRule {
    fn optimize(&self, plan: Arc<dyn ExecutionPlan>, config: &ConfigOptions) -> Result<Arc<dyn ExecutionPlan>> {
        let children = plan
            .children()
            .into_iter()
            .map(|c| self.optimize(c, config))
            .collect::<Result<Vec<_>>>()?;
        let plan = if children.is_empty() { plan } else { plan.with_new_children(children)? };
        if let Some(agg) = plan.as_any().downcast_ref::<AggregateExec>() {
            if let Some(gpu) = self.try_offload(agg)? {
                return Ok(gpu);
            }
        }
        Ok(plan)
    }
    fn name(&self) -> &str {
        "prism_gpu_offload"
    }
    fn schema_check(&self) -> bool {
        true // GpuAggExec returns exactly AggregateExec(Partial)'s schema
    }
}

/// Flattens DataFusion physical expressions into the arrays of prism_gpu_plan_desc.  The
/// CStrings are kept alive here until prism_gpu_plan_create has copied them.
#[derive(Default)]
struct DescBuilder {
    exprs: Vec<sys::prism_gpu_expr>,
    keys: Vec<sys::prism_gpu_key>,
    aggs: Vec<sys::prism_gpu_agg>,
    strings: Vec<CString>,
    literals: Vec<Vec<u8>>,
}

impl DescBuilder {
    fn push(&mut self, kind: i32, op: i32, lhs: i32, rhs: i32, column: i32, flags: i32, i64_: i64, s: Option<&[u8]>) -> i32 {
        let (ptr, len) = match s {
            Some(bytes) => {
                self.literals.push(bytes.to_vec());
                let v = self.literals.last().unwrap();
                (v.as_ptr() as *const _, v.len() as u64)
            }
            None => (std::ptr::null(), 0),
        };
        self.exprs.push(sys::prism_gpu_expr { kind, op, lhs, rhs, column, flags, i64: i64_, str: ptr, str_len: len });
        self.exprs.len() as i32 - 1
    }

    fn name(&mut self, s: &str) -> *const std::os::raw::c_char {
        self.strings.push(CString::new(s).unwrap_or_default());
        self.strings.last().unwrap().as_ptr()
    }

    fn column(&mut self, e: &Arc<dyn PhysicalExpr>) -> Option<i32> {
        e.as_any().downcast_ref::<Column>().map(|c| c.index() as i32)
    }

    /// WHERE tree; None = a shape the library does not take.
    fn expr(&mut self, e: &Arc<dyn PhysicalExpr>, schema: &SchemaRef) -> Option<i32> {
        let any = e.as_any();
        if let Some(b) = any.downcast_ref::<BinaryExpr>() {
            let op = match b.op() {
                Operator::And | Operator::Or => {
                    let l = self.expr(b.left(), schema)?;
                    let r = self.expr(b.right(), schema)?;
                    let kind = if *b.op() == Operator::And { sys::PRISM_EXPR_AND } else { sys::PRISM_EXPR_OR };
                    return Some(self.push(kind, 0, l, r, -1, 0, 0, None));
                }
                Operator::Eq => sys::PRISM_CMP_EQ,
                Operator::NotEq => sys::PRISM_CMP_NEQ,
                Operator::Lt => sys::PRISM_CMP_LT,
                Operator::LtEq => sys::PRISM_CMP_LTEQ,
                Operator::Gt => sys::PRISM_CMP_GT,
                Operator::GtEq => sys::PRISM_CMP_GTEQ,
                _ => return None,
            };
            let col = self.column(b.left())?;
            let l = self.push(sys::PRISM_EXPR_COLUMN, 0, -1, -1, col, 0, 0, None);
            let r = self.literal(b.right())?;
            return Some(self.push(sys::PRISM_EXPR_CMP, op, l, r, -1, 0, 0, None));
        }
        if let Some(l) = any.downcast_ref::<LikeExpr>() {
            let col = self.column(l.expr())?;
            let lhs = self.push(sys::PRISM_EXPR_COLUMN, 0, -1, -1, col, 0, 0, None);
            let rhs = self.literal(l.pattern())?;
            let flags = if l.negated() { sys::PRISM_LIKE_NEGATED } else { 0 }
                | if l.case_insensitive() { sys::PRISM_LIKE_CASE_INSENSITIVE } else { 0 };
            return Some(self.push(sys::PRISM_EXPR_LIKE, 0, lhs, rhs, -1, flags, 0, None));
        }
        if let Some(n) = any.downcast_ref::<NotExpr>() {
            let a = self.expr(n.arg(), schema)?;
            return Some(self.push(sys::PRISM_EXPR_NOT, 0, a, -1, -1, 0, 0, None));
        }
        if let Some(n) = any.downcast_ref::<IsNullExpr>() {
            let col = self.column(n.arg())?;
            let a = self.push(sys::PRISM_EXPR_COLUMN, 0, -1, -1, col, 0, 0, None);
            return Some(self.push(sys::PRISM_EXPR_IS_NULL, 0, a, -1, -1, 0, 0, None));
        }
        if let Some(n) = any.downcast_ref::<IsNotNullExpr>() {
            let col = self.column(n.arg())?;
            let a = self.push(sys::PRISM_EXPR_COLUMN, 0, -1, -1, col, 0, 0, None);
            return Some(self.push(sys::PRISM_EXPR_IS_NOT_NULL, 0, a, -1, -1, 0, 0, None));
        }
        None
    }

    /// Literals arrive already coerced to the column's type by DataFusion's type coercion.
    fn literal(&mut self, e: &Arc<dyn PhysicalExpr>) -> Option<i32> {
        let lit = e.as_any().downcast_ref::<Literal>()?;
        let v = match lit.value() {
            ScalarValue::Int64(Some(v)) => *v,
            ScalarValue::Int32(Some(v)) => *v as i64,
            ScalarValue::Int16(Some(v)) => *v as i64,
            ScalarValue::UInt16(Some(v)) => *v as i64,
            ScalarValue::TimestampMillisecond(Some(v), None) => *v,
            ScalarValue::Utf8(Some(s)) => return Some(self.push(sys::PRISM_EXPR_LIT_STR, 0, -1, -1, -1, 0, 0, Some(s.as_bytes()))),
            ScalarValue::Binary(Some(b)) => return Some(self.push(sys::PRISM_EXPR_LIT_STR, 0, -1, -1, -1, 0, 0, Some(b))),
            _ => return None,
        };
        Some(self.push(sys::PRISM_EXPR_LIT_I64, 0, -1, -1, -1, 0, v, None))
    }

    /// GROUP BY expression: a column, or date_bin(stride, column[, origin]) with literal stride / origin.
    fn key(&mut self, e: &Arc<dyn PhysicalExpr>, name: &str) -> bool {
        let name_ptr = self.name(name);
        if let Some(c) = e.as_any().downcast_ref::<Column>() {
            self.keys.push(sys::prism_gpu_key { kind: sys::PRISM_KEY_COLUMN, column: c.index() as i32, stride_ms: 0, origin_ms: 0, name: name_ptr });
            return true;
        }
        if let Some(f) = e.as_any().downcast_ref::<ScalarFunctionExpr>() {
            if f.name() != "date_bin" || f.args().len() < 2 {
                return false;
            }
            let stride_ms = match f.args()[0].as_any().downcast_ref::<Literal>().map(|l| l.value().clone()) {
                Some(ScalarValue::IntervalDayTime(Some(v))) => (v >> 32) * 86_400_000 + (v & 0xffff_ffff),
                Some(ScalarValue::IntervalMonthDayNano(Some(v))) if (v >> 96) == 0 => {
                    (((v >> 64) & 0xffff_ffff) as i64) * 86_400_000 + ((v & 0xffff_ffff_ffff_ffff) as i64) / 1_000_000
                }
                _ => return false,
            };
            let column = match f.args()[1].as_any().downcast_ref::<Column>() {
                Some(c) => c.index() as i32,
                None => return false,
            };
            let origin_ms = match f.args().get(2).and_then(|a| a.as_any().downcast_ref::<Literal>()).map(|l| l.value().clone()) {
                None => 0,
                Some(ScalarValue::TimestampNanosecond(Some(v), None)) => v / 1_000_000,
                Some(ScalarValue::TimestampMillisecond(Some(v), None)) => v,
                _ => return false,
            };
            if !matches!(f.return_type(), DataType::Timestamp(TimeUnit::Millisecond, None)) {
                return false;
            }
            self.keys.push(sys::prism_gpu_key { kind: sys::PRISM_KEY_DATE_BIN, column, stride_ms, origin_ms, name: name_ptr });
            return true;
        }
        false
    }

    fn agg(&mut self, a: &Arc<dyn AggregateExpr>) -> bool {
        let any = a.as_any();
        let func = if any.is::<Count>() {
            sys::PRISM_AGG_COUNT
        } else if any.is::<Sum>() {
            sys::PRISM_AGG_SUM
        } else if any.is::<Min>() {
            sys::PRISM_AGG_MIN
        } else if any.is::<Max>() {
            sys::PRISM_AGG_MAX
        } else if any.is::<Avg>() {
            sys::PRISM_AGG_AVG
        } else if any.is::<DistinctCount>() {
            sys::PRISM_AGG_COUNT_DISTINCT
        } else {
            return false;
        };
        let args = a.expressions();
        if args.len() != 1 {
            return false;
        }
        let name_ptr = self.name(a.name());
        // COUNT(*) reaches the physical plan as COUNT(UInt8(1)): a literal argument counts rows
        if func == sys::PRISM_AGG_COUNT && args[0].as_any().is::<Literal>() {
            self.aggs.push(sys::prism_gpu_agg { func: sys::PRISM_AGG_COUNT_STAR, column: -1, name: name_ptr });
            return true;
        }
        match args[0].as_any().downcast_ref::<Column>() {
            Some(c) => {
                self.aggs.push(sys::prism_gpu_agg { func, column: c.index() as i32, name: name_ptr });
                true
            }
            None => false,
        }
    }
}
