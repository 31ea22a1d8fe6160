// EXIF tag walk of the reference's camera loader (image_stitching.cpp:344-347 ExifLoader, :411-417 the ImageDescription entry) without
// libexif: the ASCII value of tag 0x010E of a JPEG file's APP1 "Exif" segment.  Row N4 of SURVEY 8(f) -- the pixel side of the
// ingest (JPEG decoding) is not rebuilt; frames reach the driver as raw BGR.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mis {
// true + *out when the file carries the tag.  The IFDs are visited in libexif's order (IFD0, IFD1, Exif sub-IFD); when several
// carry the tag the last one wins, as the reference's callback overwrites its state.  The value is cut to 1022 characters: the
// reference reads it through exif_entry_get_value into a 1024-byte buffer with maxlen 1023 (:412, :416).
bool exifImageDescription(const uint8_t* data, size_t size, std::string* out);
bool exifImageDescriptionFile(const std::string& path, std::string* out);
}  // namespace mis
