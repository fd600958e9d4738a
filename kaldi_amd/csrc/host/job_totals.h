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
// has joined.  The path must be new for every run.  The processes JOIN AT PROGRAM START, when
// all of them are alive and roughly in step (JobTotals' constructor: rendezvous and communicator
// set-up, bounded by --rendezvous-timeout); only the all-reduce itself waits until the end of the
// run, where the slices of a job finish at different times.  The totals are a log line, not a
// result: a process that cannot join, or whose all-reduce fails, logs a warning with its own
// counts and keeps the exit status of its scoring run.  A process that dies after joining
// leaves the others waiting in the all-reduce (RCCL has no time limit of its own): the job's
// time limit ends them, as it ends any other stuck process of the job.
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

  int rendezvous_timeout = 300;  // seconds the other processes wait for process 0's id at start
};

// One process's membership in the job's RCCL communicator.
class JobTotals {
 public:
  // Joins the job on GPU `device` of this process.  Never throws: Joined() says whether the
  // all-reduce is available and Error() why not.
  JobTotals(const JobTotalsOptions &opts, int device);
  ~JobTotals();
  JobTotals(const JobTotals &) = delete;
  JobTotals &operator=(const JobTotals &) = delete;
  bool Joined() const { return comm_ != nullptr; }
  const std::string &Error() const { return error_; }
  // Sums counts[0..n) over the processes, in place.  False (and Error()) when it could not.
  bool AllReduce(long long *counts, int n);

 private:
  struct Rccl;
  Rccl *rccl_ = nullptr;
  void *comm_ = nullptr;
  int device_ = 0;
  std::string error_;
};

}  // namespace kaldi_amd

#endif
