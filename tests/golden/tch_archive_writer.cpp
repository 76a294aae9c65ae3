// tch_archive_writer.cpp — writes a named-tensor archive exactly the way tch-rs does.
//
// The reference saves a network with `self.vs.save(path)` (alpha-tak/src/model/net5.rs:95-104, net6.rs likewise);
// tch 0.7's VarStore::save is Tensor::save_multi, which is torch-sys's C shim
//     at_save_multi(tensors, names, n, filename):
//         torch::serialize::OutputArchive archive;
//         for i: archive.write(std::string(names[i]), *tensors[i], /*is_buffer=*/false);
//         archive.save_to(filename);
// (torch-sys is not vendored in the reference; these are its three libtorch calls).  This program makes the same three
// calls against the libtorch that ships inside the PyTorch wheel of this image, so the resulting file has the container
// format — zip layout, pickled module with tensors as parameters, tensor storage records — of a file written by the
// reference binary.  It is test tooling: tests/golden/make_tch_archive.py feeds it and commits the output.
//
// Input (argv[1]): a flat blob  u32 count, then per tensor: u32 name_len, name bytes, u32 ndim, i64 dims[ndim], f32 data.
// Output (argv[2]): the archive.
#include <torch/serialize/output-archive.h>
#include <torch/torch.h>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s tensors.blob out.model\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) return 3;
    uint32_t count = 0;
    in.read((char*)&count, 4);
    torch::serialize::OutputArchive archive;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t len = 0, ndim = 0;
        in.read((char*)&len, 4);
        std::string name(len, '\0');
        in.read(&name[0], len);
        in.read((char*)&ndim, 4);
        std::vector<int64_t> dims(ndim);
        in.read((char*)dims.data(), 8 * ndim);
        torch::Tensor t = torch::empty(dims, torch::kFloat32);
        in.read((char*)t.data_ptr<float>(), 4 * t.numel());
        if (!in) return 4;
        archive.write(name, t, /*is_buffer=*/false);
    }
    archive.save_to(std::string(argv[2]));
    return 0;
}
