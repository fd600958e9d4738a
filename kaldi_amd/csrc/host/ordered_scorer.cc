#include "ordered_scorer.h"

#include <algorithm>
#include <deque>
#include <sstream>

namespace kaldi_amd {

OrderedScorer::OrderedScorer(const std::vector<int32> &devices, PipelineFactory make_pipeline)
    : make_pipeline_(std::move(make_pipeline)) {
  if (devices.empty()) KALDI_AMD_ERR("OrderedScorer: no device given");
  for (size_t i = 0; i < devices.size(); ++i) {
    workers_.emplace_back(new Worker);
    workers_.back()->device = devices[i];
  }
  for (size_t i = 0; i < workers_.size(); ++i)
    workers_[i]->thread = std::thread([this, i]() { Run(static_cast<int32>(i)); });
  std::string error;
  {
    std::unique_lock<std::mutex> lock(mu_);
    for (auto &w : workers_) {
      start_cv_.wait(lock, [&]() { return w->started; });
      if (error.empty() && !w->start_error.empty()) error = w->start_error;
    }
    if (!error.empty()) stop_ = true;
  }
  if (!error.empty()) {
    work_cv_.notify_all();
    for (auto &w : workers_) w->thread.join();
    workers_.clear();
    KALDI_AMD_ERR(error);
  }
}

OrderedScorer::~OrderedScorer() {
  {
    std::unique_lock<std::mutex> lock(mu_);
    stop_ = true;
  }
  work_cv_.notify_all();
  for (auto &w : workers_)
    if (w->thread.joinable()) w->thread.join();
}

void OrderedScorer::Publish(const std::vector<ScoreJob *> &window) {
  {
    std::unique_lock<std::mutex> lock(mu_);
    // Taken from the back: what is left of an older window before anything of this one (the
    // writer commits in input order and is waiting for the older jobs); within a window the
    // longest first, the earlier of two equally long ones first.
    ++windows_published_;
    for (auto it = window.rbegin(); it != window.rend(); ++it) {
      (*it)->done = false;
      (*it)->window = windows_published_;
      queue_.push_back(*it);
    }
    std::stable_sort(queue_.begin(), queue_.end(), [](const ScoreJob *a, const ScoreJob *b) {
      if (a->window != b->window) return a->window > b->window;
      return a->TotalRows() < b->TotalRows();
    });
  }
  work_cv_.notify_all();
}

void OrderedScorer::WaitFor(ScoreJob *job) {
  std::unique_lock<std::mutex> lock(mu_);
  done_cv_.wait(lock, [&]() { return job->done; });
}

ScoreJob *OrderedScorer::Take(bool block) {
  std::unique_lock<std::mutex> lock(mu_);
  if (block) work_cv_.wait(lock, [&]() { return stop_ || !queue_.empty(); });
  if (queue_.empty()) return nullptr;
  ScoreJob *job = queue_.back();
  queue_.pop_back();
  return job;
}

void OrderedScorer::Finish(ScoreJob *job) {
  {
    std::unique_lock<std::mutex> lock(mu_);
    job->done = true;
  }
  done_cv_.notify_all();
}

void OrderedScorer::ScoreAlone(LikelihoodPipeline *pipe, ScoreJob *job) {
  int32 r0 = 0, o0 = 0;
  for (size_t u = 0; u < job->keys.size(); ++u) {
    const int32 rows = job->rows[u], orows = job->out_rows[u];
    job->errors[u].clear();
    if (orows <= 0) {
      job->errors[u] = "output would be empty";
    } else {
      bool submitted = false;
      try {
        pipe->SubmitInto(job->keys[u], job->feats.RowData(r0), rows, job->feats.Stride(), job->out.RowData(o0),
                         job->out.Stride());
        submitted = true;
        pipe->WaitFront(nullptr);
      } catch (const std::exception &e) {
        job->errors[u] = e.what();
      }
      if (submitted) pipe->Pop();
    }
    r0 += rows;
    o0 += std::max(0, orows);
  }
}

void OrderedScorer::Run(int32 index) {
  Worker &w = *workers_[index];
  std::unique_ptr<LikelihoodPipeline> pipe;
  try {
    CheckAmlike(amlike_init(w.device));
    pipe = make_pipeline_();
  } catch (const std::exception &e) {
    w.start_error = e.what();
  }
  {
    std::unique_lock<std::mutex> lock(mu_);
    w.started = true;
  }
  start_cv_.notify_all();
  if (!pipe) return;

  std::deque<ScoreJob *> in_flight;
  std::vector<ScoreJob *> retry;  // failed as a whole: scored utterance by utterance once the pipeline is empty
  // the oldest job in flight: its results are complete, or it goes to `retry`
  auto complete_front = [&]() {
    ScoreJob *job = in_flight.front();
    in_flight.pop_front();
    bool ok = true;
    try {
      pipe->WaitFront(nullptr);
    } catch (const std::exception &) {
      ok = false;
    }
    pipe->Pop();
    if (ok) Finish(job);
    else retry.push_back(job);
  };
  for (;;) {
    ScoreJob *job = nullptr;
    // A job waiting to be scored again holds up the writer: nothing new is taken until it is done.
    if (!pipe->Full() && retry.empty()) job = Take(/*block=*/in_flight.empty());
    if (job) {
      job->worker = index;
      const size_t n = job->keys.size();
      job->errors.assign(n, std::string());
      job->out_rows.resize(n);
      int32 total_out = 0;
      bool all_positive = true;
      for (size_t u = 0; u < n; ++u) {
        job->out_rows[u] = pipe->OutputRows(job->rows[u]);
        all_positive = all_positive && job->out_rows[u] > 0;
        total_out += std::max(0, job->out_rows[u]);
      }
      bool submitted = false;
      try {
        job->out.Resize(total_out, pipe->NumPdfs());
        if (job->feats.NumCols() != pipe->Dim())
          KALDI_AMD_ERR("Dim mismatch: data dim = " << job->feats.NumCols() << " vs. model dim = " << pipe->Dim());
        // a batch of several utterances goes as one matrix only where rows are scored
        // independently of their neighbours (no temporal context)
        if (all_positive && (n == 1 || pipe->RowsIndependent())) {
          pipe->SubmitInto(job->keys[0], job->feats.Data(), job->feats.NumRows(), job->feats.Stride(),
                           job->out.Data(), job->out.Stride());
          submitted = true;
        }
      } catch (const std::exception &e) {
        for (size_t u = 0; u < n; ++u) job->errors[u] = e.what();
        if (job->out.NumRows() != total_out || job->feats.NumCols() != pipe->Dim()) {
          Finish(job);  // nothing to retry: no room for the result, or no utterance can fit the model
          continue;
        }
      }
      if (submitted) in_flight.push_back(job);
      else retry.push_back(job);
      continue;
    }
    if (!in_flight.empty()) {
      complete_front();
      continue;
    }
    if (!retry.empty()) {
      for (ScoreJob *r : retry) {
        ScoreAlone(pipe.get(), r);
        Finish(r);
      }
      retry.clear();
      continue;
    }
    break;  // Take(block) came back empty-handed: the scorer is shutting down
  }
}

std::vector<int32> ParseDeviceList(const std::string &list, int32 num_gpus, int32 first_device) {
  std::vector<int32> devices;
  const int32 visible = amlike_device_count();
  if (!list.empty()) {
    std::stringstream ss(list);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
      char *end = nullptr;
      const long d = strtol(tok.c_str(), &end, 10);
      if (tok.empty() || *end != '\0') KALDI_AMD_ERR("Invalid device list: " << list);
      devices.push_back(static_cast<int32>(d));
    }
  } else {
    if (num_gpus <= 0) KALDI_AMD_ERR("--num-gpus must be positive, got " << num_gpus);
    for (int32 i = 0; i < num_gpus; ++i) devices.push_back(first_device + i);
  }
  for (int32 d : devices)
    if (d < 0 || d >= visible)
      KALDI_AMD_ERR("GPU " << d << " asked for, " << visible << " visible");
  return devices;
}

}  // namespace kaldi_amd
