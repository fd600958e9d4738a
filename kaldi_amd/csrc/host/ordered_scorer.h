// Scoring on several GPUs from one process, results in input order.
//
// The reference's in-process parallel form is one task per utterance run by a pool of threads
// whose outputs are committed strictly in the order the tasks were submitted
// (/root/reference/src/thread/kaldi-task-sequence.h:78-129, used by
// /root/reference/src/gmmbin/gmm-latgen-faster-parallel.cc:138-155 and
// /root/reference/src/nnet2bin/nnet-latgen-faster-parallel.cc).  Here the pool is one worker
// thread per GPU, each with its own copy of the model on its device and two jobs in flight on
// it; the main thread reads jobs (a job = one utterance, or a batch of whole utterances), hands
// them over a window at a time, and gets them back in input order.
//
// Within a window the workers take the longest job first (the tail of a window is then made of
// short jobs, and the GPUs finish it together).  A job that fails (an exception of the scoring
// call: non-finite scores, dimension mismatch ...) is retried utterance by utterance; the
// utterances that fail alone are marked, the others keep their results: one bad utterance costs
// itself, not the run (the reference's drivers count it and go on,
// /root/reference/src/gmmbin/gmm-latgen-faster.cc:118-122).
#ifndef KALDI_AMD_CSRC_HOST_ORDERED_SCORER_H_
#define KALDI_AMD_CSRC_HOST_ORDERED_SCORER_H_

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pipeline.h"

namespace kaldi_amd {

// One unit of work: whole utterances, their feature rows stacked in `feats`.
struct ScoreJob {
  std::vector<std::string> keys;
  std::vector<int32> rows;  // feature rows of each utterance (> 0)
  Matrix feats;             // [sum of rows x dim]
  // filled in by the scorer
  PinnedMatrix out;                  // [sum of out_rows x num_pdfs]
  std::vector<int32> out_rows;       // output rows of each utterance
  std::vector<std::string> errors;   // per utterance: empty = scored
  int32 worker = -1;                 // which worker scored it
  bool done = false;
  long long window = 0;              // which Publish() call handed it over
  int32 TotalRows() const { return feats.NumRows(); }
};

class OrderedScorer {
 public:
  // Called on a worker's thread once its device is current: builds that worker's pipeline
  // (its own handle on that device).
  using PipelineFactory = std::function<std::unique_ptr<LikelihoodPipeline>()>;
  // One worker per entry of `devices` (a device may appear several times: several workers
  // sharing one GPU, each with its own handle).  Throws if a worker cannot start.
  OrderedScorer(const std::vector<int32> &devices, PipelineFactory make_pipeline);
  ~OrderedScorer();
  int32 NumWorkers() const { return static_cast<int32>(workers_.size()); }
  // The number of jobs one window should hold (two per worker).
  int32 WindowSize() const { return 2 * NumWorkers(); }
  // Hands a window of jobs to the workers (they stay owned by the caller, who must keep them
  // alive and untouched until WaitFor has returned for each).
  void Publish(const std::vector<ScoreJob *> &window);
  // Blocks until `job` is done.
  void WaitFor(ScoreJob *job);

 private:
  struct Worker {
    std::thread thread;
    int32 device = 0;
    std::string start_error;
    bool started = false;
  };
  void Run(int32 index);
  void ScoreAlone(LikelihoodPipeline *pipe, ScoreJob *job);  // utterance by utterance
  ScoreJob *Take(bool block);
  void Finish(ScoreJob *job);

  PipelineFactory make_pipeline_;
  std::vector<std::unique_ptr<Worker>> workers_;
  std::mutex mu_;
  std::condition_variable work_cv_, done_cv_, start_cv_;
  std::vector<ScoreJob *> queue_;  // taken from the back: oldest window last, within a window longest last
  long long windows_published_ = 0;
  bool stop_ = false;
};

// "0,1,2" -> {0, 1, 2}; with an empty list: devices 0 .. num_gpus-1.  Checks them against the
// visible devices.
std::vector<int32> ParseDeviceList(const std::string &list, int32 num_gpus, int32 first_device);

}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_ORDERED_SCORER_H_
