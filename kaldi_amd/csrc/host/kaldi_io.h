// Kaldi object I/O for the host layer: stream primitives, Vector / Matrix containers and
// their text + binary forms, archive (ark / scp) readers and writers.
//
// Formats follow the reference (paths relative to /root/reference/src):
//   base/io-funcs.cc:26-204, base/io-funcs-inl.h:32-215   tokens, ints, floats, bool, "\0B"
//   matrix/kaldi-vector.cc:1107-1244                      Vector  ("FV"/"DV", text "[ .. ]")
//   matrix/kaldi-matrix.cc:1118-1352                      Matrix  ("FM"/"DM", text rows)
//   util/kaldi-holder-inl.h:45-100, util/kaldi-table.h    ark / scp tables
// Errors are std::runtime_error, as KALDI_ERR throws (base/kaldi-error.cc:173-185).
#ifndef KALDI_AMD_CSRC_HOST_KALDI_IO_H_
#define KALDI_AMD_CSRC_HOST_KALDI_IO_H_

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace kaldi_amd {

typedef float BaseFloat;
typedef int32_t int32;

[[noreturn]] void KaldiErr(const std::string &msg);
#define KALDI_AMD_ERR(stream_expr)          \
  do {                                      \
    std::ostringstream kaldi_amd_os_;       \
    kaldi_amd_os_ << stream_expr;           \
    ::kaldi_amd::KaldiErr(kaldi_amd_os_.str()); \
  } while (0)

// ---- stream primitives -----------------------------------------------------
bool InitKaldiInputStream(std::istream &is, bool *binary);  // detects "\0B"
void InitKaldiOutputStream(std::ostream &os, bool binary);
void ReadToken(std::istream &is, bool binary, std::string *tok);
void ExpectToken(std::istream &is, bool binary, const char *tok);
// Accepts either "<tok1> <tok2>" or just "<tok2>" (nnet2/nnet-component.cc ExpectOneOrTwoTokens).
void ExpectOneOrTwoTokens(std::istream &is, bool binary, const std::string &tok1,
                          const std::string &tok2);
void WriteToken(std::ostream &os, bool binary, const std::string &tok);
void ReadBasicType(std::istream &is, bool binary, int32 *v);
void ReadBasicType(std::istream &is, bool binary, float *v);   // accepts a stored double
void ReadBasicType(std::istream &is, bool binary, double *v);  // accepts a stored float
void ReadBasicType(std::istream &is, bool binary, bool *v);
void WriteBasicType(std::ostream &os, bool binary, int32 v);
void WriteBasicType(std::ostream &os, bool binary, float v);
void WriteBasicType(std::ostream &os, bool binary, double v);
void WriteBasicType(std::ostream &os, bool binary, bool v);
void ReadIntegerVector(std::istream &is, bool binary, std::vector<int32> *v);
void WriteIntegerVector(std::ostream &os, bool binary, const std::vector<int32> &v);

// ---- containers ------------------------------------------------------------
class Vector {
 public:
  Vector() {}
  explicit Vector(int32 dim) : data_(dim, 0.0f) {}
  int32 Dim() const { return static_cast<int32>(data_.size()); }
  void Resize(int32 dim) { data_.assign(dim, 0.0f); }
  float *Data() { return data_.data(); }
  const float *Data() const { return data_.data(); }
  float &operator()(int32 i) { return data_[i]; }
  float operator()(int32 i) const { return data_[i]; }
  void Read(std::istream &is, bool binary);  // float or double on disk
  void Write(std::ostream &os, bool binary) const;

 private:
  std::vector<float> data_;
};

// Row-major, rows padded to a multiple of 4 floats (16 bytes) like kaldi::Matrix.
class Matrix {
 public:
  Matrix() {}
  Matrix(int32 rows, int32 cols) { Resize(rows, cols); }
  void Resize(int32 rows, int32 cols);
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  int32 Stride() const { return stride_; }
  float *Data() { return data_.data(); }
  const float *Data() const { return data_.data(); }
  float *RowData(int32 r) { return data_.data() + static_cast<size_t>(r) * stride_; }
  const float *RowData(int32 r) const { return data_.data() + static_cast<size_t>(r) * stride_; }
  float &operator()(int32 r, int32 c) { return RowData(r)[c]; }
  float operator()(int32 r, int32 c) const { return RowData(r)[c]; }
  void Swap(Matrix *other);
  void Read(std::istream &is, bool binary);  // "FM", "DM", "CM" / "CM2" (compressed) or text
  void Write(std::ostream &os, bool binary) const;

 private:
  void ReadCompressed(std::istream &is, int format);
  int32 rows_ = 0, cols_ = 0, stride_ = 0;
  std::vector<float> data_;
};

// ---- files -------------------------------------------------------------------
// A shell command as a stream, as the reference's piped rxfilenames / wxfilenames
// ("gunzip -c feats.gz |", "| gzip -c > likes.gz"; util/kaldi-io.cc PipeInputImpl /
// PipeOutputImpl): popen() behind a std::streambuf.  A non-zero exit status of the command
// is reported on stderr when the stream is closed, as the reference does.
class PipeStream {
 public:
  PipeStream(const std::string &command, bool for_reading);
  ~PipeStream();
  std::istream &In() { return *in_; }
  std::ostream &Out() { return *out_; }

 private:
  class Buf;
  std::string command_;
  FILE *fp_ = nullptr;
  std::unique_ptr<Buf> buf_;
  std::unique_ptr<std::istream> in_;
  std::unique_ptr<std::ostream> out_;
};
// True for "cmd |" (input) / "| cmd" (output); *command gets the command without the bar.
bool IsInputPipe(const std::string &rxfilename, std::string *command);
bool IsOutputPipe(const std::string &wxfilename, std::string *command);

// rxfilename: a path, "-" (stdin), "path:offset" or "command |".  Opens and detects the
// binary marker.
class Input {
 public:
  Input(const std::string &rxfilename, bool *binary);
  std::istream &Stream() { return *is_; }

 private:
  std::unique_ptr<std::ifstream> file_;
  std::unique_ptr<PipeStream> pipe_;
  std::istream *is_ = nullptr;
};

class Output {
 public:
  Output(const std::string &wxfilename, bool binary, bool write_header = true);
  ~Output();
  std::ostream &Stream() { return *os_; }

 private:
  std::unique_ptr<std::ofstream> file_;
  std::unique_ptr<PipeStream> pipe_;
  std::ostream *os_ = nullptr;
};

// ---- tables of Matrix<float> ---------------------------------------------------
// rspecifier: "ark:FILE", "ark:-", "scp:FILE" (options such as ",s,cs,t" are accepted and
// ignored on read).  Mirrors SequentialBaseFloatMatrixReader (util/table-types.h:36-74).
class SequentialMatrixReader {
 public:
  explicit SequentialMatrixReader(const std::string &rspecifier);
  bool Done() const { return done_; }
  const std::string &Key() const { return key_; }
  const Matrix &Value() const { return value_; }
  void Next();

 private:
  void ReadOne();
  bool is_scp_ = false;
  std::unique_ptr<std::ifstream> file_;
  std::unique_ptr<PipeStream> pipe_;
  std::istream *is_ = nullptr;
  bool done_ = false;
  std::string key_;
  Matrix value_;
};

// wspecifier: "ark:FILE", "ark,t:FILE", "ark:-", "ark,scp:ARK,SCP".  Mirrors BaseFloatMatrixWriter.
void WriteMatrixData(std::ostream &os, bool binary, const float *data, int32 rows, int32 cols,
                     int32 stride);

class MatrixWriter {
 public:
  explicit MatrixWriter(const std::string &wspecifier);
  ~MatrixWriter();
  void Write(const std::string &key, const Matrix &m);
  // rows given by pointer and stride (no intermediate Matrix)
  void Write(const std::string &key, const float *data, int32 rows, int32 cols, int32 stride);

 private:
  bool binary_ = true;
  std::string ark_name_;
  std::unique_ptr<std::ofstream> file_, scp_;
  std::unique_ptr<PipeStream> pipe_;
  std::ostream *os_ = nullptr;
};

}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_KALDI_IO_H_
