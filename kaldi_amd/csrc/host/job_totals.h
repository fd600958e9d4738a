// End-of-run totals over the processes of a split job, one process per GPU.
//
// The reference runs a data set as JOB=1:nj processes, each with its own slice of the
// feature table and its own output (/root/reference/egs/wsj/s5/steps/decode.sh, `$cmd
// JOB=1:$nj`), and every process logs its own counts (gmmbin/gmm-compute-likes.cc:81-83,
// nnet2bin/nnet-am-compute.cc:127-128).  With one such process per GPU of a node, the
// counters {utterances done, utterances failed, frames} are also summed over the processes
// with one RCCL all-reduce at the end of the run (over xGMI; nothing is exchanged while
// scoring), and each process logs the totals of the whole job beside its own.
//
// Rendezvous: process 0 writes RCCL's unique id to `rendezvous_path` (written under another
// name, then renamed), the others wait for the file; process 0 removes it once every process
// has joined.  The path must be new for every run.  A process that dies before the end of its
// run leaves the others waiting for it in the communicator set-up (RCCL has no time limit of its
// own): the job's time limit ends them, as it ends any other stuck process of the job.
#ifndef KALDI_AMD_CSRC_HOST_JOB_TOTALS_H_
#define KALDI_AMD_CSRC_HOST_JOB_TOTALS_H_

#include <string>

namespace kaldi_amd {

struct JobTotalsOptions {
  int num_procs = 1;  // processes of the job
  int proc_id = 0;    // this one, 0-based
  std::string rendezvous;
  // Reduction requested: more than one process, or a rendezvous path given (a job of one
  // process then runs the same RCCL calls).
  bool Enabled() const { return num_procs > 1 || !rendezvous.empty(); }
};

// Sums counts[0..n) over the processes, in place, on GPU `device` of this process.  Throws
// std::runtime_error when RCCL cannot be loaded, the rendezvous times out (`timeout_s`) or a
// RCCL / HIP call fails.
void AllReduceJobTotals(const JobTotalsOptions &opts, int device, long long *counts, int n, int timeout_s = 300);

}  // namespace kaldi_amd

#endif
