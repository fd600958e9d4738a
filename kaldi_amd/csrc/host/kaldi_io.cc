// Kaldi object I/O for the host layer (see kaldi_io.h for the format references).
#include "kaldi_io.h"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace kaldi_amd {

void KaldiErr(const std::string &msg) { throw std::runtime_error("ERROR (kaldi_amd) " + msg); }

bool InitKaldiInputStream(std::istream &is, bool *binary) {
  if (is.peek() == '\0') {
    is.get();
    if (is.peek() != 'B') return false;
    is.get();
    *binary = true;
  } else {
    *binary = false;
  }
  return true;
}

void InitKaldiOutputStream(std::ostream &os, bool binary) {
  if (binary) {
    os.put('\0');
    os.put('B');
  }
  if (os.precision() < 7) os.precision(7);
}

void ReadToken(std::istream &is, bool binary, std::string *tok) {
  if (!binary) is >> std::ws;
  is >> *tok;
  if (is.fail()) KALDI_AMD_ERR("ReadToken, failed to read token at file position " << is.tellg());
  if (!isspace(is.peek()))
    KALDI_AMD_ERR("ReadToken, expected space after token, saw instead "
                  << static_cast<char>(is.peek()) << ", at file position " << is.tellg());
  is.get();  // the single separator
}

void ExpectToken(std::istream &is, bool binary, const char *tok) {
  std::string got;
  const std::streampos pos = is.tellg();
  if (!binary) is >> std::ws;
  is >> got;
  if (is.fail() || got != tok)
    KALDI_AMD_ERR("Failed to read token [started at file position " << pos << "], expected " << tok
                                                                     << " (got " << got << ")");
  is.get();
}

void ExpectOneOrTwoTokens(std::istream &is, bool binary, const std::string &tok1,
                          const std::string &tok2) {
  std::string t;
  ReadToken(is, binary, &t);
  if (t == tok1) {
    ExpectToken(is, binary, tok2.c_str());
  } else if (t != tok2) {
    KALDI_AMD_ERR("Expecting token " << tok1 << " or " << tok2 << " but got " << t);
  }
}

void WriteToken(std::ostream &os, bool binary, const std::string &tok) {
  (void)binary;
  os << tok << " ";
  if (os.fail()) KaldiErr("Write failure in WriteToken.");
}

void ReadBasicType(std::istream &is, bool binary, int32 *v) {
  if (binary) {
    const int len = is.get();
    if (len != 4) KALDI_AMD_ERR("ReadBasicType: did not get expected integer type, " << len << " vs. 4");
    is.read(reinterpret_cast<char *>(v), 4);
  } else {
    is >> *v;
  }
  if (is.fail()) KALDI_AMD_ERR("Read failure in ReadBasicType, file position is " << is.tellg());
}

static void ReadFloating(std::istream &is, bool binary, double *v) {
  if (binary) {
    const int c = is.peek();
    if (c == 4) {
      is.get();
      float f;
      is.read(reinterpret_cast<char *>(&f), 4);
      *v = f;
    } else if (c == 8) {
      is.get();
      is.read(reinterpret_cast<char *>(v), 8);
    } else {
      KALDI_AMD_ERR("ReadBasicType: expected float, saw " << c << ", at file position " << is.tellg());
    }
  } else {
    std::string s;
    is >> std::ws >> s;
    char *end = nullptr;
    *v = strtod(s.c_str(), &end);
    if (s.empty() || *end != '\0') KALDI_AMD_ERR("ReadBasicType: bad number \"" << s << "\"");
  }
  if (is.fail()) KALDI_AMD_ERR("ReadBasicType: failed to read, at file position " << is.tellg());
}

void ReadBasicType(std::istream &is, bool binary, float *v) {
  double d;
  ReadFloating(is, binary, &d);
  *v = static_cast<float>(d);
}
void ReadBasicType(std::istream &is, bool binary, double *v) { ReadFloating(is, binary, v); }

void ReadBasicType(std::istream &is, bool binary, bool *v) {
  if (!binary) is >> std::ws;
  const int c = is.get();
  if (c == 'T') *v = true;
  else if (c == 'F') *v = false;
  else KALDI_AMD_ERR("Read failure in ReadBasicType<bool>, saw " << static_cast<char>(c));
}

void WriteBasicType(std::ostream &os, bool binary, int32 v) {
  if (binary) {
    os.put(4);
    os.write(reinterpret_cast<const char *>(&v), 4);
  } else {
    os << v << " ";
  }
}
void WriteBasicType(std::ostream &os, bool binary, float v) {
  if (binary) {
    os.put(4);
    os.write(reinterpret_cast<const char *>(&v), 4);
  } else {
    os << v << " ";
  }
}
void WriteBasicType(std::ostream &os, bool binary, double v) {
  if (binary) {
    os.put(8);
    os.write(reinterpret_cast<const char *>(&v), 8);
  } else {
    os << v << " ";
  }
}
void WriteBasicType(std::ostream &os, bool binary, bool v) {
  os << (v ? "T" : "F");
  if (!binary) os << " ";
}

void ReadIntegerVector(std::istream &is, bool binary, std::vector<int32> *v) {
  if (binary) {
    const int sz = is.peek();
    if (sz != 4) KALDI_AMD_ERR("ReadIntegerVector: expected to see type of size 4, saw instead " << sz);
    is.get();
    int32 n;
    is.read(reinterpret_cast<char *>(&n), 4);
    if (is.fail() || n < 0) KaldiErr("ReadIntegerVector: read failure");
    v->resize(n);
    if (n > 0) is.read(reinterpret_cast<char *>(v->data()), static_cast<std::streamsize>(4) * n);
  } else {
    v->clear();
    is >> std::ws;
    if (is.peek() != '[') KaldiErr("ReadIntegerVector: expected to see [");
    is.get();
    is >> std::ws;
    while (is.peek() != ']') {
      int32 x;
      is >> x >> std::ws;
      if (is.fail()) KaldiErr("ReadIntegerVector: read failure");
      v->push_back(x);
    }
    is.get();
  }
  if (is.fail()) KaldiErr("ReadIntegerVector: read failure");
}

void WriteIntegerVector(std::ostream &os, bool binary, const std::vector<int32> &v) {
  if (binary) {
    os.put(4);
    const int32 n = static_cast<int32>(v.size());
    os.write(reinterpret_cast<const char *>(&n), 4);
    if (n) os.write(reinterpret_cast<const char *>(v.data()), static_cast<std::streamsize>(4) * n);
  } else {
    os << "[ ";
    for (int32 x : v) os << x << " ";
    os << "]\n";
  }
}

// ---- Vector / Matrix ---------------------------------------------------------
static float ParseFloatToken(const std::string &s) {
  char *end = nullptr;
  const float f = strtof(s.c_str(), &end);  // accepts inf / nan spellings
  if (s.empty() || *end != '\0') KALDI_AMD_ERR("Failed to read matrix or vector from stream: bad number \"" << s << "\"");
  return f;
}

void Vector::Read(std::istream &is, bool binary) {
  if (binary) {
    std::string tok;
    ReadToken(is, binary, &tok);
    int32 dim;
    if (tok == "FV") {
      ReadBasicType(is, binary, &dim);
      data_.resize(dim);
      if (dim) is.read(reinterpret_cast<char *>(data_.data()), static_cast<std::streamsize>(4) * dim);
    } else if (tok == "DV") {
      ReadBasicType(is, binary, &dim);
      std::vector<double> tmp(dim);
      if (dim) is.read(reinterpret_cast<char *>(tmp.data()), static_cast<std::streamsize>(8) * dim);
      data_.assign(tmp.begin(), tmp.end());
    } else {
      KALDI_AMD_ERR("Failed to read vector from stream: expected token FV or DV, got " << tok);
    }
    if (is.fail()) KaldiErr("Failed to read vector from stream: read error");
  } else {
    std::string s;
    is >> s;
    if (is.fail() || s != "[") KALDI_AMD_ERR("Failed to read vector from stream: expected \"[\" but got " << s);
    data_.clear();
    while (true) {
      is >> s;
      if (is.fail()) KaldiErr("Failed to read vector from stream: unexpected end");
      if (s == "]") break;
      data_.push_back(ParseFloatToken(s));
    }
    // the reference consumes the rest of the line
    if (is.peek() == '\r') is.get();
    if (is.peek() == '\n') is.get();
  }
}

void Vector::Write(std::ostream &os, bool binary) const {
  if (binary) {
    WriteToken(os, binary, "FV");
    WriteBasicType(os, binary, Dim());
    if (Dim()) os.write(reinterpret_cast<const char *>(data_.data()), static_cast<std::streamsize>(4) * Dim());
  } else {
    os << " [ ";
    for (float x : data_) os << x << " ";
    os << "]\n";
  }
  if (os.fail()) KaldiErr("Failed to write vector to stream");
}

void Matrix::Resize(int32 rows, int32 cols) {
  rows_ = rows;
  cols_ = cols;
  stride_ = (cols + 3) & ~3;
  data_.assign(static_cast<size_t>(rows) * stride_, 0.0f);
}

void Matrix::Swap(Matrix *o) {
  std::swap(rows_, o->rows_);
  std::swap(cols_, o->cols_);
  std::swap(stride_, o->stride_);
  data_.swap(o->data_);
}

// CompressedMatrix on disk (matrix/compressed-matrix.cc:404-470): after the token, a header
// {float min_value, float range, int32 num_rows, int32 num_cols}; then
//   "CM"  (format 1): num_cols x {uint16 p0, p25, p75, p100}, then num_cols columns of
//         num_rows bytes; a byte maps piecewise-linearly between the column's percentiles
//         (CharToFloat :363-373);
//   "CM2" (format 2): num_rows x num_cols uint16, row-major.
// A uint16 u stands for min_value + range * u / 65535 (Uint16ToFloat :244-250).
void Matrix::ReadCompressed(std::istream &is, int format) {
  struct {
    float min_value, range;
    int32 num_rows, num_cols;
  } h;
  is.read(reinterpret_cast<char *>(&h), sizeof(h));
  if (is.fail()) KaldiErr("Failed to read header");
  if (h.num_cols == 0) {
    Resize(0, 0);
    return;
  }
  if (h.num_rows < 0 || h.num_cols < 0) KaldiErr("Failed to read matrix from stream: negative size");
  Resize(h.num_rows, h.num_cols);
  auto u16 = [&](uint16_t v) { return h.min_value + h.range * 1.52590218966964e-05F * v; };
  if (format == 1) {
    std::vector<uint16_t> pc(static_cast<size_t>(4) * h.num_cols);
    is.read(reinterpret_cast<char *>(pc.data()), static_cast<std::streamsize>(pc.size() * 2));
    std::vector<unsigned char> col(h.num_rows);
    for (int32 c = 0; c < h.num_cols; ++c) {
      const float p0 = u16(pc[4 * c]), p25 = u16(pc[4 * c + 1]), p75 = u16(pc[4 * c + 2]),
                  p100 = u16(pc[4 * c + 3]);
      is.read(reinterpret_cast<char *>(col.data()), h.num_rows);
      for (int32 r = 0; r < h.num_rows; ++r) {
        const unsigned char v = col[r];
        float f;
        if (v <= 64) f = p0 + (p25 - p0) * v * (1 / 64.0);
        else if (v <= 192) f = p25 + (p75 - p25) * (v - 64) * (1 / 128.0);
        else f = p75 + (p100 - p75) * (v - 192) * (1 / 63.0);
        (*this)(r, c) = f;
      }
    }
  } else {
    std::vector<uint16_t> row(h.num_cols);
    for (int32 r = 0; r < h.num_rows; ++r) {
      is.read(reinterpret_cast<char *>(row.data()), static_cast<std::streamsize>(2) * h.num_cols);
      for (int32 c = 0; c < h.num_cols; ++c) (*this)(r, c) = u16(row[c]);
    }
  }
  if (is.fail()) KaldiErr("Failed to read data.");
}

void Matrix::Read(std::istream &is, bool binary) {
  if (binary) {
    std::string tok;
    ReadToken(is, binary, &tok);
    if (tok == "CM" || tok == "CM2") {
      ReadCompressed(is, tok == "CM" ? 1 : 2);
      return;
    }
    if (tok != "FM" && tok != "DM")
      KALDI_AMD_ERR("Failed to read matrix from stream: expected token FM or DM, got " << tok);
    int32 rows, cols;
    ReadBasicType(is, binary, &rows);
    ReadBasicType(is, binary, &cols);
    if (rows < 0 || cols < 0) KaldiErr("Failed to read matrix from stream: negative size");
    Resize(rows, cols);
    if (tok == "FM") {
      for (int32 r = 0; r < rows; ++r)
        is.read(reinterpret_cast<char *>(RowData(r)), static_cast<std::streamsize>(4) * cols);
    } else {
      std::vector<double> tmp(cols);
      for (int32 r = 0; r < rows; ++r) {
        is.read(reinterpret_cast<char *>(tmp.data()), static_cast<std::streamsize>(8) * cols);
        for (int32 c = 0; c < cols; ++c) RowData(r)[c] = static_cast<float>(tmp[c]);
      }
    }
    if (is.fail()) KaldiErr("Failed to read matrix from stream: read error");
    return;
  }
  // Text: "[", rows ended by newline or ';', "]" (matrix/kaldi-matrix.cc:1260-1350).
  std::string s;
  is >> s;
  if (is.fail() || s != "[") KALDI_AMD_ERR("Failed to read matrix from stream: expected \"[\", got \"" << s << "\"");
  std::vector<std::vector<float>> rows;
  std::vector<float> cur;
  while (true) {
    const int c = is.peek();
    if (c == EOF) KaldiErr("Failed to read matrix from stream: end of file");
    if (c == '\n' || c == ';') {
      is.get();
      if (!cur.empty()) {
        rows.push_back(cur);
        cur.clear();
      }
    } else if (c == ' ' || c == '\t' || c == '\r') {
      is.get();
    } else if (c == ']') {
      is.get();
      if (!cur.empty()) rows.push_back(cur);
      if (is.peek() == '\r') is.get();
      if (is.peek() == '\n') is.get();
      break;
    } else {
      is >> s;
      if (!s.empty() && s.back() == ']') {  // "1.0]" without a space
        s.pop_back();
        if (!s.empty()) cur.push_back(ParseFloatToken(s));
        if (!cur.empty()) rows.push_back(cur);
        break;
      }
      cur.push_back(ParseFloatToken(s));
    }
  }
  const int32 nr = static_cast<int32>(rows.size()), nc = nr ? static_cast<int32>(rows[0].size()) : 0;
  Resize(nr, nc);
  for (int32 r = 0; r < nr; ++r) {
    if (static_cast<int32>(rows[r].size()) != nc)
      KaldiErr("Failed to read matrix from stream: inconsistent #columns");
    memcpy(RowData(r), rows[r].data(), sizeof(float) * nc);
  }
}

// matrix/kaldi-matrix.cc:1118-1144 for rows given by pointer and stride.
void WriteMatrixData(std::ostream &os, bool binary, const float *data, int32 rows, int32 cols,
                     int32 stride) {
  if (binary) {
    WriteToken(os, binary, "FM");
    WriteBasicType(os, binary, rows);
    WriteBasicType(os, binary, cols);
    if (stride == cols) {
      os.write(reinterpret_cast<const char *>(data), static_cast<std::streamsize>(4) * cols * rows);
    } else {
      for (int32 r = 0; r < rows; ++r)
        os.write(reinterpret_cast<const char *>(data + static_cast<size_t>(r) * stride),
                 static_cast<std::streamsize>(4) * cols);
    }
  } else {
    if (cols == 0) {
      os << " [ ]\n";
    } else {
      os << " [";
      for (int32 r = 0; r < rows; ++r) {
        os << "\n  ";
        for (int32 c = 0; c < cols; ++c) os << data[static_cast<size_t>(r) * stride + c] << " ";
      }
      os << "]\n";
    }
  }
  if (os.fail()) KaldiErr("Failed to write matrix to stream");
}

void Matrix::Write(std::ostream &os, bool binary) const {
  WriteMatrixData(os, binary, data_.data(), rows_, cols_, stride_);
}

// ---- files ---------------------------------------------------------------------
static void SplitOffset(const std::string &rx, std::string *path, std::streamoff *off) {
  *path = rx;
  *off = -1;
  const size_t pos = rx.find_last_of(':');
  if (pos != std::string::npos && pos + 1 < rx.size() &&
      rx.find_first_not_of("0123456789", pos + 1) == std::string::npos) {
    *path = rx.substr(0, pos);
    *off = static_cast<std::streamoff>(strtoll(rx.c_str() + pos + 1, nullptr, 10));
  }
}

// ---- pipes -------------------------------------------------------------------
class PipeStream::Buf : public std::streambuf {
 public:
  Buf(FILE *fp, bool reading) : fp_(fp), buf_(1 << 16) {
    if (reading) setg(buf_.data(), buf_.data(), buf_.data());
    else setp(buf_.data(), buf_.data() + buf_.size());
  }

 protected:
  int_type underflow() override {
    if (gptr() < egptr()) return traits_type::to_int_type(*gptr());
    const size_t n = fread(buf_.data(), 1, buf_.size(), fp_);
    if (n == 0) return traits_type::eof();
    setg(buf_.data(), buf_.data(), buf_.data() + n);
    return traits_type::to_int_type(*gptr());
  }
  int_type overflow(int_type ch) override {
    if (sync() != 0) return traits_type::eof();
    if (!traits_type::eq_int_type(ch, traits_type::eof())) {
      *pptr() = traits_type::to_char_type(ch);
      pbump(1);
    }
    return traits_type::not_eof(ch);
  }
  int sync() override {
    if (!pbase()) return 0;
    const size_t n = static_cast<size_t>(pptr() - pbase());
    if (n && fwrite(pbase(), 1, n, fp_) != n) return -1;
    setp(buf_.data(), buf_.data() + buf_.size());
    return fflush(fp_) == 0 ? 0 : -1;
  }

 private:
  FILE *fp_;
  std::vector<char> buf_;
};

PipeStream::PipeStream(const std::string &command, bool for_reading) : command_(command) {
  fp_ = popen(command.c_str(), for_reading ? "r" : "w");
  if (!fp_) KALDI_AMD_ERR("Failed opening pipe for " << (for_reading ? "reading" : "writing")
                                                    << ", command is: " << command);
  buf_.reset(new Buf(fp_, for_reading));
  if (for_reading) in_.reset(new std::istream(buf_.get()));
  else out_.reset(new std::ostream(buf_.get()));
}

PipeStream::~PipeStream() {
  if (out_) out_->flush();
  in_.reset();
  out_.reset();
  buf_.reset();
  if (fp_) {
    const int status = pclose(fp_);
    if (status) std::cerr << "WARNING (kaldi_amd) Pipe " << command_ << " had nonzero return status " << status << "\n";
  }
}

static std::string Trim(const std::string &s) {
  const size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
  return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
}

bool IsInputPipe(const std::string &rxfilename, std::string *command) {
  const std::string t = Trim(rxfilename);
  if (t.size() < 2 || t.back() != '|') return false;
  *command = Trim(t.substr(0, t.size() - 1));
  return !command->empty();
}

bool IsOutputPipe(const std::string &wxfilename, std::string *command) {
  const std::string t = Trim(wxfilename);
  if (t.size() < 2 || t.front() != '|') return false;
  *command = Trim(t.substr(1));
  return !command->empty();
}

Input::Input(const std::string &rxfilename, bool *binary) {
  std::string command;
  if (rxfilename == "-" || rxfilename.empty()) {
    is_ = &std::cin;
  } else if (IsInputPipe(rxfilename, &command)) {
    pipe_.reset(new PipeStream(command, true));
    is_ = &pipe_->In();
  } else {
    std::string path;
    std::streamoff off;
    SplitOffset(rxfilename, &path, &off);
    file_.reset(new std::ifstream(path, std::ios::in | std::ios::binary));
    if (!file_->is_open()) KALDI_AMD_ERR("Error opening input stream " << rxfilename);
    if (off >= 0) file_->seekg(off);
    is_ = file_.get();
  }
  bool b = false;
  if (!InitKaldiInputStream(*is_, &b)) KALDI_AMD_ERR("Error initializing Kaldi input stream " << rxfilename);
  if (binary) *binary = b;
}

Output::Output(const std::string &wxfilename, bool binary, bool write_header) {
  std::string command;
  if (wxfilename == "-" || wxfilename.empty()) {
    os_ = &std::cout;
  } else if (IsOutputPipe(wxfilename, &command)) {
    pipe_.reset(new PipeStream(command, false));
    os_ = &pipe_->Out();
  } else {
    file_.reset(new std::ofstream(wxfilename, std::ios::out | std::ios::binary));
    if (!file_->is_open()) KALDI_AMD_ERR("Error opening output stream " << wxfilename);
    os_ = file_.get();
  }
  if (write_header) InitKaldiOutputStream(*os_, binary);
}

Output::~Output() {
  if (os_) os_->flush();
}

// ---- tables ----------------------------------------------------------------------
static void SplitSpecifier(const std::string &spec, std::string *opts, std::string *rest) {
  const size_t pos = spec.find(':');
  if (pos == std::string::npos) KALDI_AMD_ERR("Invalid rspecifier/wspecifier " << spec);
  *opts = spec.substr(0, pos);
  *rest = spec.substr(pos + 1);
}

static bool HasOpt(const std::string &opts, const std::string &o) {
  std::stringstream ss(opts);
  std::string item;
  while (std::getline(ss, item, ',')) if (item == o) return true;
  return false;
}

SequentialMatrixReader::SequentialMatrixReader(const std::string &rspecifier) {
  std::string opts, rest;
  SplitSpecifier(rspecifier, &opts, &rest);
  if (HasOpt(opts, "scp")) is_scp_ = true;
  else if (!HasOpt(opts, "ark")) KALDI_AMD_ERR("Invalid rspecifier " << rspecifier);
  std::string command;
  if (rest == "-" || rest.empty()) {
    is_ = &std::cin;
  } else if (IsInputPipe(rest, &command)) {
    pipe_.reset(new PipeStream(command, true));
    is_ = &pipe_->In();
  } else {
    file_.reset(new std::ifstream(rest, std::ios::in | std::ios::binary));
    if (!file_->is_open()) KALDI_AMD_ERR("Error opening table " << rspecifier);
    is_ = file_.get();
  }
  ReadOne();
}

void SequentialMatrixReader::Next() { ReadOne(); }

void SequentialMatrixReader::ReadOne() {
  *is_ >> std::ws;
  if (is_->eof() || is_->peek() == EOF) {
    done_ = true;
    return;
  }
  *is_ >> key_;
  if (is_->fail()) {
    done_ = true;
    return;
  }
  if (is_scp_) {
    std::string rx;
    std::getline(*is_, rx);
    const size_t b = rx.find_first_not_of(" \t"), e = rx.find_last_not_of(" \t\r");
    if (b == std::string::npos) KALDI_AMD_ERR("Invalid scp line for key " << key_);
    bool binary;
    Input in(rx.substr(b, e - b + 1), &binary);
    value_.Read(in.Stream(), binary);
  } else {
    const int c = is_->get();  // the space after the key
    if (c != ' ' && c != '\t') KALDI_AMD_ERR("Invalid archive: expected space after key " << key_);
    bool binary;
    if (!InitKaldiInputStream(*is_, &binary)) KALDI_AMD_ERR("Invalid archive entry for key " << key_);
    value_.Read(*is_, binary);
  }
}

MatrixWriter::MatrixWriter(const std::string &wspecifier) {
  std::string opts, rest;
  SplitSpecifier(wspecifier, &opts, &rest);
  if (!HasOpt(opts, "ark")) KALDI_AMD_ERR("Invalid wspecifier " << wspecifier);
  binary_ = !HasOpt(opts, "t");
  std::string scp_name;
  if (HasOpt(opts, "scp")) {
    const size_t comma = rest.find(',');
    if (comma == std::string::npos) KALDI_AMD_ERR("Invalid wspecifier " << wspecifier);
    scp_name = rest.substr(comma + 1);
    rest = rest.substr(0, comma);
  }
  ark_name_ = rest;
  std::string command;
  if (rest == "-" || rest.empty()) {
    os_ = &std::cout;
  } else if (IsOutputPipe(rest, &command)) {
    if (!scp_name.empty()) KALDI_AMD_ERR("Invalid wspecifier (an scp cannot point into a pipe) " << wspecifier);
    pipe_.reset(new PipeStream(command, false));
    os_ = &pipe_->Out();
  } else {
    file_.reset(new std::ofstream(rest, std::ios::out | std::ios::binary));
    if (!file_->is_open()) KALDI_AMD_ERR("Error opening table for writing " << wspecifier);
    os_ = file_.get();
  }
  if (!scp_name.empty()) {
    scp_.reset(new std::ofstream(scp_name));
    if (!scp_->is_open()) KALDI_AMD_ERR("Error opening scp for writing " << scp_name);
  }
}

MatrixWriter::~MatrixWriter() {
  if (os_) os_->flush();
}

void MatrixWriter::Write(const std::string &key, const Matrix &m) {
  Write(key, m.Data(), m.NumRows(), m.NumCols(), m.Stride());
}

void MatrixWriter::Write(const std::string &key, const float *data, int32 rows, int32 cols,
                         int32 stride) {
  *os_ << key << ' ';
  if (scp_) *scp_ << key << ' ' << ark_name_ << ':' << os_->tellp() << '\n';
  InitKaldiOutputStream(*os_, binary_);
  WriteMatrixData(*os_, binary_, data, rows, cols, stride);
  if (os_->fail()) KALDI_AMD_ERR("Write failure for key " << key);
}

}  // namespace kaldi_amd
