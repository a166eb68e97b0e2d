// Step plans: one training step recorded once as the list of its launches and replayed by one C call.
//
// The reference runs model.fit(batch_size=1, steps_per_epoch=180) (model_training.py:299): the same static schedule 180
// times.  Here a step is ~250 kernel launches on two HIP streams; issued from the Python schedule they cost the host
// ~1.5 ms (ctypes marshalling, launch-plan selection, event bookkeeping) per step.  A plan keeps, per launch, the kernel,
// its grid, its stream and a copy of its arguments (lisec::launch, common.h) plus the event records / waits between the
// two streams, and lisec_step_plan_run re-issues them in order: plain hipLaunchKernel calls, no graph (a captured HIP graph
// of this two-stream step replays 2x slower than the eager launches on ROCm 7.2).
// What varies from step to step must live in device memory the recorded launches point at (the sweep's points in a
// fixed-capacity buffer, the targets, the optimizer's iteration counter): arguments are frozen at record time.
#include "common.h"

#include <vector>

namespace lisec {

struct StepPlan {
    std::vector<std::function<hipError_t()>> ops;
    bool recording = false;
};

namespace {
thread_local StepPlan* t_recording = nullptr;
}

StepPlan* plan_recording() { return t_recording; }
void plan_append(StepPlan* plan, std::function<hipError_t()> op) { plan->ops.push_back(std::move(op)); }

}  // namespace lisec

using namespace lisec;

extern "C" int lisec_step_plan_create(lisec_step_plan_t* plan) {
    LISEC_CHECK_ARG(plan, "NULL plan");
    *plan = new StepPlan();
    return LISEC_OK;
}

extern "C" int lisec_step_plan_destroy(lisec_step_plan_t plan) {
    StepPlan* p = static_cast<StepPlan*>(plan);
    if (p && t_recording == p) t_recording = nullptr;
    delete p;
    return LISEC_OK;
}

extern "C" int lisec_step_plan_begin(lisec_step_plan_t plan) {
    StepPlan* p = static_cast<StepPlan*>(plan);
    LISEC_CHECK_ARG(p, "NULL plan");
    LISEC_CHECK_ARG(!t_recording, "this thread is already recording a step plan");
    p->ops.clear();
    p->recording = true;
    t_recording = p;
    return LISEC_OK;
}

extern "C" int lisec_step_plan_end(lisec_step_plan_t plan) {
    StepPlan* p = static_cast<StepPlan*>(plan);
    LISEC_CHECK_ARG(p && t_recording == p, "this thread is not recording that plan");
    p->recording = false;
    t_recording = nullptr;
    return LISEC_OK;
}

extern "C" int lisec_step_plan_recording(void) { return t_recording ? 1 : 0; }

extern "C" int lisec_step_plan_size(lisec_step_plan_t plan) {
    StepPlan* p = static_cast<StepPlan*>(plan);
    return p ? (int)p->ops.size() : -1;
}

extern "C" int lisec_step_plan_run(lisec_step_plan_t plan) {
    StepPlan* p = static_cast<StepPlan*>(plan);
    LISEC_CHECK_ARG(p && !p->recording, "NULL plan, or a plan that is still recording");
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const hipError_t e = p->ops[i]();
        if (e != hipSuccess) {
            set_error("step plan: operation %zu of %zu failed: %s", i, p->ops.size(), hipGetErrorString(e));
            return LISEC_EHIP;
        }
    }
    return LISEC_OK;
}

// Event record / wait between the streams of a step, through the library so that a recording plan sees them.
extern "C" int lisec_event_record(void* event, lisec_stream_t stream) {
    LISEC_CHECK_ARG(event, "NULL event");
    hipEvent_t ev = static_cast<hipEvent_t>(event);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (StepPlan* p = plan_recording()) plan_append(p, [=]() { return hipEventRecord(ev, st); });
    LISEC_HIP_TRY(hipEventRecord(ev, st));
    return LISEC_OK;
}

extern "C" int lisec_stream_wait_event(lisec_stream_t stream, void* event) {
    LISEC_CHECK_ARG(event, "NULL event");
    hipEvent_t ev = static_cast<hipEvent_t>(event);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (StepPlan* p = plan_recording()) plan_append(p, [=]() { return hipStreamWaitEvent(st, ev, 0); });
    LISEC_HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    return LISEC_OK;
}

// A host function as a step of the plan: executed now, and -- when the calling thread is recording -- re-executed by
// lisec_step_plan_run at the same place in the sequence.  For work that a step needs between its launches and that is
// not a launch of this library: a gradient exchange issued through another runtime (torch.distributed over gloo), a
// host-side hand-off.  fn returns 0 on success; any other value stops the run with LISEC_EHIP.
extern "C" int lisec_step_plan_host_call(int (*fn)(void*), void* arg) {
    LISEC_CHECK_ARG(fn, "NULL host function");
    if (StepPlan* p = plan_recording()) plan_append(p, [=]() { return fn(arg) == 0 ? hipSuccess : hipErrorUnknown; });
    if (fn(arg) != 0) {
        set_error("step plan: host call failed");
        return LISEC_EHIP;
    }
    return LISEC_OK;
}
